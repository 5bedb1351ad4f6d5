"""Deprecated alias of modules.norm, kept because the reference keeps it (src/norm.py:3)."""
import warnings

from modules.norm import *  # noqa: F401,F403

warnings.warn("Use modules.norm instead.", FutureWarning)
