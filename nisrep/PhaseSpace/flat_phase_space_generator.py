from nf_amd.PhaseSpace.flat_phase_space_generator import *  # noqa: F401,F403
from nf_amd.PhaseSpace.flat_phase_space_generator import FlatInvertiblePhasespace, PhaseSpaceGeneratorError, VirtualPhaseSpaceGenerator  # noqa: F401
