"""`adhoc_driver` of the Conv-TasNet recipe (egs/wsj0-mix/conv-tasnet/src/adhoc_driver.py), served by speech_amd.driver."""
import _ctn_alias  # noqa: F401
from speech_amd.driver import AdhocTrainer, TesterBase


class Tester(TesterBase):
    pass
