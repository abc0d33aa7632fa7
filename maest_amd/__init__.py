"""maest_amd: MI355X-native (gfx950) implementation of the MAEST mel -> patchout-ViT hot path,
behind the reference's Python surface (``from maest import get_maest`` -> ``from maest_amd import get_maest``)."""
import os as _os

# Hardware queues.  The HIP runtime multiplexes a process's streams onto GPU_MAX_HW_QUEUES hardware queues per device (default 4), and a
# queue runs its packets in order: a stream that waits on an event holds up every other stream of its queue.  A data-parallel training
# process has the caller's stream, the engine's weight-gradient and exchange streams (maest.py: _engine_stream) and RCCL's own -- with four
# queues they share, and the one-rank forced-collective step measured 53.5 ms against 45.7 without the exchange path; with eight queues
# 47.0 (profiles/r06_hw_queues.txt).  The runtime reads the variable when it initialises, i.e. at the process's first HIP call: set here
# unless the job already chose a value; import maest_amd (or set it in the job's environment) before touching the device.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .maest import MAEST, get_maest  # noqa: F401,E402

__all__ = ["get_maest", "MAEST"]
