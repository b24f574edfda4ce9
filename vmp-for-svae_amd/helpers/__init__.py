from . import logging_utils, scheduling, tf_utils  # noqa: F401
