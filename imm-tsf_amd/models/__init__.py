"""Drop-in `models` package (same import paths and Model(args).forecasting(...) signatures as the reference)."""
