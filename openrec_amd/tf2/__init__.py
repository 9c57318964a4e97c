"""Mirror of the reference's `openrec.tf2` package surface:
same module / class names, constructor arguments and call signatures, executed
by libopenrec_hip.so instead of TensorFlow."""
from . import modules, recommenders  # noqa: F401
