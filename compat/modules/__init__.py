"""Top-level `modules` package of the reference (src/modules/), served by speech_amd.modules."""
from _ctn_alias import expose

expose(__name__, "speech_amd.modules")
