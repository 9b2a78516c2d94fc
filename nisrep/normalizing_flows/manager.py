import torch  # noqa: F401  (the reference module re-exports torch through `import *`)
import numpy as np  # noqa: F401
from nf_amd.normalizing_flows.manager import *  # noqa: F401,F403
from nf_amd.normalizing_flows.manager import (AffineManager, BasicManager, ModelAPI, PWLinManager, PWQuadManager,  # noqa: F401
                                              get_bin, normal)
from nf_amd.normalizing_flows.layers.layers import *  # noqa: F401,F403
from nf_amd.normalizing_flows.layers.coupling_cells import AffineCoupling, PWLin, PWQuad  # noqa: F401
