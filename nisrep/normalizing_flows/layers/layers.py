from nf_amd.normalizing_flows.layers.layers import *  # noqa: F401,F403
