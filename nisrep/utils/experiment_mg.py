from nf_amd.utils.experiment_mg import *  # noqa: F401,F403
from nf_amd.utils.experiment_mg import epoch_length, pro  # noqa: F401
