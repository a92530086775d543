from .fusion import *  # noqa: F401,F403
from .pooling import *  # noqa: F401,F403
from .dropout import *  # noqa: F401,F403
from .modules import UnimodalBranch, IdentityBranch  # noqa: F401
