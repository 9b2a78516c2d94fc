"""Import-path compatibility shim: `from nisrep.normalizing_flows.manager import *` and
`from nisrep.PhaseSpace.flat_phase_space_generator import *` (reference README.md:33,51) resolve to nf_amd."""
