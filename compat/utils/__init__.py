"""Top-level `utils` package of the reference (src/utils/), served by speech_amd.utils."""
from _ctn_alias import expose

expose(__name__, "speech_amd.utils")
