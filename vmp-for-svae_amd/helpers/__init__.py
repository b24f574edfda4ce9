from . import tf_utils  # noqa: F401
