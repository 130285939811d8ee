"""The reference's public surface (/root/reference/lib/index.ts:1-12), same names."""
from .ac import formatAcResult, simulateAC  # noqa: F401
from .netlist import parseNetlist  # noqa: F401
from .simulate import (eecEngineTranToVGraphs, formatTranResult, simulate, simulateTRAN,  # noqa: F401
                       spiceyTranToVGraphs)

__all__ = ["parseNetlist", "simulate", "simulateAC", "simulateTRAN", "formatAcResult", "formatTranResult",
           "spiceyTranToVGraphs", "eecEngineTranToVGraphs"]
