"""Drop-in `fusions` package: same module/class names, constructor and forward signatures and state_dict keys as
the reference's fusions/ directory, computed by the HIP kernels in libimmtsf_hip.so."""
