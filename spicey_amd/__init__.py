"""spicey_amd — MI355X-native drop-in for tscircuit/spicey's simulation paths.

`spicey_amd.api` carries the public names of /root/reference/lib/index.ts:1-12 (parseNetlist, simulate, simulateAC,
simulateTRAN, formatAcResult, formatTranResult, spiceyTranToVGraphs, eecEngineTranToVGraphs).  The solvers (transient
and AC) run in libspicey_hip.so (include/spicey_hip.h); nothing here computes on the CPU.  Importing this package does
not load the library: spicey_amd.lib does on first use, and fails loudly when it is missing.
"""
