from nf_amd.normalizing_flows.layers import *  # noqa
