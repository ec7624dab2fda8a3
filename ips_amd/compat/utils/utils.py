from ips_amd.utils.utils import *  # noqa: F401,F403
from ips_amd.utils.utils import Logger, Struct, adjust_learning_rate, shuffle_batch, shuffle_instance  # noqa: F401
