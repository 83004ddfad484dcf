"""tendermintx_amd -- MI355X-native witness generator for the TendermintX skip / step circuits.

Only what the hot path needs: the C-ABI library (csrc/ -> libtmx.so, hand-written HIP for gfx950) and a thin
host-side mirror of the reference's interface for this path (circuits.py).  There is no CPU implementation in
this package; everything fails loudly without libtmx.so and a HIP device."""
from ._lib import FLAG_PRESENT, FLAG_SIGNED, KIND_SKIP, KIND_STEP, TmxError  # noqa: F401
from .context import Context  # noqa: F401
from .circuits import (CELESTIA_CHAIN_ID_BYTES, MOCHA_4_CHAIN_ID_BYTES, SKIP_MAX, InputDataFetcher, SkipCircuit,  # noqa: F401
                       StepCircuit)
