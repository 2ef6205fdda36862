

import os as _os

# Kernel arguments in device memory (HIP runtime option): every launch of the ~400 kernels of a step starts a little
# sooner (measured: C2 step -4 %, metric step -1 %).  Only effective when set before the HIP runtime initialises, i.e.
# when this package (or the entry script) is imported before the first GPU call; harmless otherwise.
_os.environ.setdefault("HIP_FORCE_DEV_KERNARG", "1")
# Hardware queues the process's HIP streams share (default 4): the compute stream, the weight-gradient side stream, c10d's communication
# stream and RCCL's own streams need one each, or launches that should overlap serialise (profiles/r05_hwq_matrix.txt).  Same condition.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
