import torch  # noqa: F401  (the reference module re-exports torch through `import *`)
import numpy as np  # noqa: F401
from nf_amd.normalizing_flows.manager import *  # noqa: F401,F403
from nf_amd.normalizing_flows.manager import (BasicManager, ModelAPI, PWLinManager, PWQuadManager, get_bin,  # noqa: F401
                                              normal)
from nf_amd.normalizing_flows.layers.layers import *  # noqa: F401,F403
from nf_amd.normalizing_flows.layers.coupling_cells import PWLin, PWQuad  # noqa: F401
