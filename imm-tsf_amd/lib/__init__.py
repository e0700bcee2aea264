"""Drop-in `lib` package: the pieces of the reference's lib/ that sit on the fusion hot path (loss / train step)."""
