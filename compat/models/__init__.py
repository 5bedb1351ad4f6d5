"""Top-level `models` package of the reference (src/models/), served by speech_amd.models."""
from _ctn_alias import expose

expose(__name__, "speech_amd.models")
