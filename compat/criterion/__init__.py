"""Top-level `criterion` package of the reference (src/criterion/), served by speech_amd.criterion."""
from _ctn_alias import expose

expose(__name__, "speech_amd.criterion")
