"""Deprecated alias of modules.conv, kept because the reference keeps it (src/conv.py:3)."""
import warnings

from modules.conv import *  # noqa: F401,F403

warnings.warn("Use modules.conv instead.", FutureWarning)
