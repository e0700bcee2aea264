"""Process-wide knobs of the HIP path."""
import os
import threading

_PRECISIONS = {"fp32": 0, "bf16": 1}

# "fp32": exact-fp32 MFMA (parity with the reference's fp32 CPU path); "bf16": bf16 MFMA operands, fp32 accumulate.
precision = os.environ.get("IMMTSF_PRECISION", "fp32")

# How the reference's `torch.isnan(x).any()` guards (fusions/FusionModel.py:103-112, TTF_*.py:116/75) are honoured:
#   "sync"     -- like the reference: check (and host-sync) inside forward, raise ValueError immediately (5 host syncs per step: the
#                 seam then runs every launch eagerly, 3.2 ms per step at the benchmark configuration)
#   "deferred" -- (default since round 5; a documented deviation) the same ValueError, raised at the NEXT host-visible point: the kernels
#                 OR a device flag, lib.evaluation leaves "loss is NaN / notes held NaN" in pinned memory by an asynchronous copy, and the
#                 next compute_all_losses() / evaluation() call -- or FusionModel.check_nan() -- raises; no host sync inside the step
#   "off"      -- no checks
nan_check = os.environ.get("IMMTSF_NAN_CHECK", "deferred")


# While set, backward passes whose parameter gradients go to FlatTrainer sinks only compute the DATA gradients and queue
# the parameter-gradient work in immtsf.ops (run_deferred() enqueues it): immtsf.train.PhasedStep uses it to take that
# work off the path between the loss and the backbone's backward.  Off everywhere else.
defer_param_grads = False


def precision_code(p=None) -> int:
    p = precision if p is None else p
    if p not in _PRECISIONS:
        raise ValueError(f"precision must be one of {sorted(_PRECISIONS)}, got {p!r}")
    return _PRECISIONS[p]


# dropout seeds: (base seed, call counter) -> one 64-bit Philox key per forward call
_lock = threading.Lock()
_base_seed = None
_counter = 0


def manual_seed(seed: int):
    global _base_seed, _counter
    with _lock:
        _base_seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        _counter = 0


def next_seed() -> int:
    """A fresh Philox key; deterministic after torch.manual_seed()/immtsf.config.manual_seed()."""
    global _base_seed, _counter
    with _lock:
        if _base_seed is None:
            import torch
            _base_seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
        _counter += 1
        x = (_base_seed + 0x9E3779B97F4A7C15 * _counter) & 0xFFFFFFFFFFFFFFFF
        # splitmix64 finaliser
        x ^= x >> 30
        x = (x * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 27
        x = (x * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
        x ^= x >> 31
        return x


# ---- device-side step counters (hipGraph replay) -----------------------------------------------------------
# When a training step is captured into a hipGraph every host-side scalar is frozen into the graph.  Two scalars
# must still change per replay: the dropout key and Adam's step number.  With device counters enabled the fusion
# kernels add *dropout_step to their Philox key and the fused Adam kernel reads/increments *adam_step.
_device_counters = {}


def enable_device_counters(device):
    """-> (adam_step int64[1], dropout_step int64[1]) tensors on `device`; from now on every fusion forward on that
    device passes the dropout counter to the kernels."""
    import torch
    key = str(device)
    if key not in _device_counters:
        _device_counters[key] = (torch.zeros(1, dtype=torch.int64, device=device),
                                 torch.zeros(1, dtype=torch.int64, device=device))
    return _device_counters[key]


def disable_device_counters(device=None):
    if device is None:
        _device_counters.clear()
    else:
        _device_counters.pop(str(device), None)


def dropout_counter_ptr(device):
    c = _device_counters.get(str(device))
    return None if c is None else c[1].data_ptr()

# Alternative formulations kept as cross-checks of the default ones (tests flip these attributes; no environment variables):
# MMF_XAttn_Add in its low-rank form (csrc/xrank.hip) wherever its limits allow; False: the full-rank key/value + query halves
xattn_rank = True
# ... and, for a training step whose loss is the masked MSE with known observation counts, the Q half + loss + their backward as ONE
# launch (MMF_XAttn_Add.forward_loss); False: the Q half and the loss as separate ops
xattn_fused_loss = True
# FusionModel: TTF_T2V_XAttn's proj_out composed into MMF_XAttn_Add's low-rank projection: "auto" / True = wherever the pair of blocks
# allows it, False = the two blocks as written (tests run both)
fuse_tail = "auto"
# the zero-edit seam (lib.evaluation.compute_all_losses) serves a repeated (model, fusion, batch shape) from a replayed hipGraph when
# nan_check is "deferred" (no host syncs inside the step); False: every call is launched eagerly
seam_graph = True
# TTF_T2V_XAttn: "auto" = the library chooses by batch size (the folded form, csrc/t2v_fold.hip, from IMMTSF_T2V_FOLD_MIN_ROWS padded note
# rows on -- its parameter-only chains are a fixed cost; its mix-first variant, csrc/t2v_premix.hip, for windows of more than 64 padded
# notes in bf16 mode), "fold" = the folded form wherever its limits hold, "mix" = the mix-first variant wherever ITS limits hold (bf16
# mode, one head, T <= 32), "chain" = the reference's GEMM chain as written
t2v_form = "auto"
# immtsf.train.FlagStep <-> MMFXRankQLossFn: address of the device flag that says "dY_ts is ready" (None: nobody is waiting)
head_done_flag = None
head_dy_ptr = None        # ... and, when a head took the flag: the address of the dY_ts buffer its kernel publishes
# TTF_T2V_XAttn on PackedNotes: use the batch's prebuilt ragged index (PackedNotes.index(), built once per batch) instead of deriving
# it inside every forward; False: the call derives it (two launches at its head) -- the cross-check
note_index = True
# the fused TTF_T2V_XAttn -> MMF_XAttn_Add tail (FusionModel.fused_tail) hands Z over as its bf16 image ALONE (no fp32 Z: it has no reader
# in the bf16 dataflow) and takes the gradient back in LOW-RANK form (dP, Wc) -- dZ = dP Wc is formed inside the LayerNorm backward
# instead of being written and re-read (0.8 GB per step at 4096 windows, one launch on the text chain at 64).  False: the dense hand-over
# in both directions -- the cross-check
# MMF_GR_Add in split form (csrc/gr_train.hip: the text columns of its two linear maps ahead of the backbone, the Y half + loss + backward as
# one launch); False: the block as written -- the cross-check
gr_split = True
z_handover = os.environ.get("IMMTSF_Z_HANDOVER", "1") != "0"
# FullAttention over <= 32 positions with heads up to 256 wide as one kernel per direction (csrc/attn_mid.hip); False: batched GEMMs +
# row softmax
attn_mid = True
# a torch.cuda.Stream on which MMF_XAttn_Add's fold (parameters only) may run ahead of the text side (None: in line); the stream
# must be ordered behind the previous optimizer step (immtsf.train.FlagStep forks it at the start of the captured step)
fold_stream = None
param_tail = None        # FlagStep: {"flag": (address, time-out report address), "jobs": []} -- ops whose parameter-gradient tail nothing but the
                         # optimizer waits for (MMF_XAttn_Add's chain rule through the fold) set the flag behind the data half and
                         # leave the tail as a job (a callable taking a raw stream) for the branch that has time to spare
sched_gate = None        # GraphedStep: (flag address, time-out report address) -- TTF_T2V_XAttn's backward sets the flag behind its row-bound
                         # kernels, the patch encoder's backward (parameter gradients only, on the backbone's stream) spins on it first
sched_armed = None       # stream of the TTF_T2V_XAttn forward that will set it (None: nobody will -- nobody may wait)
fold_flag = None          # (flag address, time-out report address): hand the fold over through a device flag instead of a stream event
