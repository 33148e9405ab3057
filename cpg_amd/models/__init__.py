"""Model zoo of the CPG hot path (reference: models/__init__.py)."""
from . import layers  # noqa: F401
from .resnet import *  # noqa: F401,F403
from .spherenet import *  # noqa: F401,F403
from .vgg import *  # noqa: F401,F403
