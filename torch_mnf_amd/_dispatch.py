"""Which kernel family a layer call takes: every shape / row-count threshold of the Python layer in one place.

The C library picks among its kernels by SHAPE (a specialised kernel where the shape has one, else the run-time-shaped
matrix-core kernels of csrc/mnf_rt.h, else the VALU any-shape kernels); what is decided here is by ROW COUNT -- where a
faster-per-row kernel does not pay for its extra launches yet -- plus the test / measurement switches.  INTEGRATION.md
("Which kernel runs") is this table in prose; tools/coverage_map.py measures it; tests set these attributes directly
(``monkeypatch.setattr(torch_mnf_amd._dispatch, "NSF_PAD_MIN_ROWS", 0)``).  Environment switches of the package, all of
them: MNF_LIB_PATH (another build of the library), MNF_DETERMINISTIC (fixed-order gradient sums; read by the library),
MNF_FP32_MFMA (fp32 instead of split-f16 matrix-core arithmetic), MNF_CHECK_PARAMS (stale-image detector),
MNF_NO_RUN_FUSION and MNF_NO_PAIR_FUSION (layer-by-layer passes, for per-layer measurements).

| layer (direction)        | rows            | shape                                             | kernel family                  |
|--------------------------|-----------------|---------------------------------------------------|--------------------------------|
| AffineHalfFlow fwd       | any             | 3 hidden layers <= 32 (64 at d = 32/64/128), d <= 256 | ahf_split(_stack) / ahf_mfma |
|                          | >= RT_MIN_ROWS  | any h_sizes (>= 1 layer, widths 4..256), any d    | ahf_rt                         |
|                          | else            | anything                                          | ahf_generic (VALU)             |
| AffineHalfFlow bwd       | >= BWD_SPLIT_MIN_ROWS | the split kernel's shapes                   | ahf_bwd_split                  |
|                          | any             | the fp32-MFMA kernel's shapes                     | ahf_bwd_mfma_fp32              |
|                          | >= RT_MIN_ROWS  | 1..4 hidden layers of widths 4..64, any d         | ahf_bwd_rt                     |
|                          | else            | anything                                          | ahf_bwd_generic                |
| NSF_CL fwd               | any             | d % 8 == 0 up to 64, n_h <= 16, K 5/8 (10: d<=32) | nsf_mfma_split                 |
|                          | >= NSF_PAD_MIN_ROWS | other d <= 64 (zero-padded twin layer)        | nsf_mfma_split                 |
|                          | >= RT_MIN_ROWS  | any d, K 2..16, hidden widths 4..64               | nsf_rt                         |
|                          | else            | anything                                          | nsf_generic                    |
| NSF_CL bwd               | any             | the tile kernel's shapes                          | nsf_bwd_tile (+ fix-up)        |
|                          | >= RT_MIN_ROWS  | any d, K 2..16, 1..4 hidden layers of widths 4..64 | nsf_bwd_rt                    |
|                          | else            | anything                                          | nsf_bwd_generic                |
| RNVP fwd                 | few (C side)    | one hidden layer <= 64                            | rnvp_few                       |
|                          | any             | one hidden layer <= 64, d >= 49                   | rnvp_resident / narrow / split |
|                          | >= RT_MIN_ROWS  | any number of layers of widths 4..256, any d      | rnvp_rt                        |
|                          | else            | anything                                          | rnvp_generic                   |
| RNVP bwd                 | few             | one hidden layer <= 64                            | rnvp_bwd_few                   |
|                          | not rnvp_bwd_small() | one hidden layer <= 64, padded d >= 64       | rnvp_bwd_mfma                  |
|                          | >= RT_MIN_ROWS  | 1..4 layers of widths 4..128, any d               | rnvp_bwd_rt                    |
|                          | else            | anything                                          | rnvp_bwd_generic               |

Two requests override the shape: an fp32 request (layer.force_fp32_mfma / MNF_FP32_MFMA=1) never lands on the *_rt
kernels, whose arithmetic is split-f16 -- it takes the fp32 matrix-core kernel where the shape has one (AffineHalfFlow and
RNVP: wherever a split kernel exists; NSF_CL: the K = 8 shapes, plain layer only), else the VALU kernel; and under
MNF_DETERMINISTIC=1 the *_bwd_rt kernels (atomic sums) refuse, so their shapes take the VALU gradient kernels.
layer.force_generic = 1 / 2 forces the VALU / the run-time-shaped kernels (tests, tools/coverage_map.py).
"""

# The run-time-shaped matrix-core kernels (csrc/mnf_rt.h: any layer count and widths) take a call without a per-shape
# kernel from this many rows on; below, the VALU any-shape kernels (a workgroup per few rows) have the lower latency.
# csrc/mnf_host.h kRtMinRows is the same number for the forward entry points.
RT_MIN_ROWS = 2048

# AffineHalfFlow gradients: the split-f16 kernel needs three small launches more per backward pass (gradient scale,
# operand repack, fix-up list) than the fp32-MFMA one and only pays them back from ~32k rows on (4,096 rows: 0.84 vs 0.65
# ms per 9-layer training step; 32,768: 0.72 vs 0.65; 65,536: 0.73 vs 0.75; 2^20: 4.7 vs 6.5)
BWD_SPLIT_MIN_ROWS = 49152
BWD_FP32 = False  # measurements: AffineHalfFlow gradients on the fp32-MFMA kernel at every row count

# NSF_CL with halves that are not whole float4 groups (dim = 2, 6, 10, ...) runs the per-shape matrix-core kernels on a
# zero-padded twin layer from this many rows on (NSF_CL._run_padded; tools/time_nsf_padded_twin.py, dim = 2, K = 8,
# n_h = 16, forward + backward, twin vs any-shape kernels: 690 vs 416 us at 16,384 rows, 705 vs 1,119 at 65,536)
NSF_PAD_MIN_ROWS = 49152
NSF_BWD_KERNEL = "tile"  # "generic": tests / measurements run the VALU gradient kernel where the tile kernel exists

# RNVP gradients.  The per-shape matrix-core pass (four launches) from these rows / dims on (d = 800: 227 vs 252 us at 128
# rows, 284 vs 837 us at 2,048; d = 100: 326 vs 403 us at 4,096 rows; d = 50 / 64 level at 16,384 rows, 457 vs 562 / 659
# at 32,768: tools/time_rnvp_bwd_small_dim.py)
RNVP_BWD_MFMA_MIN_ROWS = 64
RNVP_BWD_MFMA_MIN_DIM = 128
RNVP_BWD_MFMA_MID_DIM, RNVP_BWD_MFMA_MID_ROWS, RNVP_BWD_MFMA_ANY_DIM_ROWS = 96, 4096, 24576
RNVP_KEEP_Y_MIN_ROWS = 4096    # the forward pass keeps y = net(mask z) for the gradient pass from this many rows on
RNVP_BWD_GENERIC = False       # measurements: the VALU gradient kernel
RNVP_BWD_FEW_GRID_OFF = False  # tests: the matrix-core / VALU gradient kernels at every row count

NO_FUSED_LOGPROB = False  # measurements: the log-prob epilogue stays its own launch after an affine run


def rnvp_bwd_small(rows: int, dim: int) -> bool:
    """True where the per-shape matrix-core RNVP gradient pass does not pay yet (few rows, or a narrow layer at a moderate
    batch): the run-time-shaped kernel (from RT_MIN_ROWS rows on) or the VALU kernel takes the call."""
    if rows < RNVP_BWD_MFMA_MIN_ROWS:
        return True
    if dim >= RNVP_BWD_MFMA_MIN_DIM or rows >= RNVP_BWD_MFMA_ANY_DIM_ROWS:
        return False
    return not (dim >= RNVP_BWD_MFMA_MID_DIM and rows >= RNVP_BWD_MFMA_MID_ROWS)


# ---------------------------------------------------------------------------------------------------------------------
# The table above as a function: which TIER a call lands on ("per-shape", "rt" = run-time-shaped, "valu").  The shape
# questions go to the library's own host-side queries (no GPU needed); the row-count questions are the constants above.
# tests/test_abi_symbols.py checks it against every row of the committed coverage map (profiles/r6/coverage_map.txt):
# the table, this function and the measured map cannot drift apart unnoticed.
# ---------------------------------------------------------------------------------------------------------------------
def _rt_range(kind: str, direction: str, hidden, K) -> bool:
    if not hidden or min(hidden) < 4:
        return False
    if kind == "ahf":
        return max(hidden) <= (256 if direction == "fwd" else 64) and (direction == "fwd" or len(hidden) <= 4)
    if kind == "nsf":
        return max(hidden) <= 64 and K is not None and 2 <= K <= 16 and (direction == "fwd" or len(hidden) <= 4)
    return max(hidden) <= (256 if direction == "fwd" else 128) and (direction == "fwd" or len(hidden) <= 4)


def tier(kind: str, direction: str, rows: int, dim: int, hidden, K: int | None = None, scale: bool = True,
         shift: bool = True) -> str:
    """Tier of one layer call in default mode (no force_generic, no fp32 request, no MNF_DETERMINISTIC): kind "ahf" |
    "nsf" | "rnvp", direction "fwd" | "bwd", hidden = the conditioner's hidden widths (NSF_CL: (n_h,) * 3)."""
    from . import _lib
    lib, hid, n = _lib.load(), _lib.int_array(list(hidden)), len(hidden)
    if kind == "ahf":
        if direction == "fwd":
            per_shape = lib.mnf_affine_half_image_floats(dim, n, hid, int(scale), int(shift)) > 0
        else:  # the fp32-MFMA gradient kernel's shapes contain the split kernel's
            per_shape = lib.mnf_affine_half_bwd_index_ints(dim, n, hid, int(scale), int(shift)) > 0
    elif kind == "nsf":
        tile = bool(lib.mnf_nsf_cl_bwd_tile_supported(dim, K, n, hid))
        per_shape = lib.mnf_nsf_cl_image_floats(dim, K, n, hid) > 0 if direction == "fwd" else tile
        hp = (dim // 2 + 3) // 4 * 4
        if not per_shape and hp != dim // 2 and 2 * hp <= 64 and len(set(hidden)) == 1 and rows >= NSF_PAD_MIN_ROWS:
            per_shape = bool(lib.mnf_nsf_cl_bwd_tile_supported(2 * hp, K, n, hid))  # the zero-padded twin layer
    else:
        per_shape = lib.mnf_rnvp_image_floats(dim, n, hid) > 0
        if direction == "bwd":
            per_shape = per_shape and not rnvp_bwd_small(rows, dim)
    if per_shape:
        return "per-shape"
    return "rt" if rows >= RT_MIN_ROWS and _rt_range(kind, direction, hidden, K) else "valu"


def tier_of_kernel(name: str) -> str:
    """The tier a kernel family name (torch_mnf_amd.last_kernel()) belongs to."""
    return "valu" if "generic" in name else "rt" if name.endswith("_rt") else "per-shape"
