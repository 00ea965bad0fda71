"""Which kernel family a layer call takes: every shape / row-count threshold of the Python layer in one place
("Which kernel runs", is this table in prose; tools/coverage_map.py measures it)."""

# The run-time-shaped matrix-core kernels (csrc/mnf_rt.h: any layer count and widths) take a call without a per-shape
# kernel from this many rows on; below, the VALU any-shape kernels (a workgroup per few rows) have the lower latency.
# csrc/mnf_host.h kRtMinRows is the same number for the forward entry points.
RT_MIN_ROWS = 2048
