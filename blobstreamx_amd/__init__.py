"""MI355X-native witness engine for Blobstream X `header_range` (host side; the kernels live in csrc/, the C ABI in
include/bsx.h)."""
import os

# The pipelined engine (engine.PipelinedEngines) drives 2 main + 2 side HIP streams beside the default stream, the
# library's own stream and RCCL's.  HIP maps streams onto GPU_MAX_HW_QUEUES hardware queues (default 4) and streams that
# share a queue serialise: measured 3.2 ms/step -> 4.8 ms/step for a second engine set at 8 queues, no effect at 16.
# The variable is read when the HIP runtime initialises, i.e. at the first device call after this import.
os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
