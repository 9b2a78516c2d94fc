from nf_amd.normalizing_flows import *  # noqa
