"""MI355X-native GANtron training engine."""
import os as _os

# The step uses several HIP streams (main, auxiliary decoder chain, weight-gradient leaf, loader copies) and RCCL
# adds its own; the runtime default of 4 hardware queues per process makes them time-slice (+7 % per step measured
# once RCCL is initialised).  Only effective when this package is imported before the first HIP call.
_os.environ.setdefault('GPU_MAX_HW_QUEUES', '8')
