"""`driver` of the wsj0-mix recipes (egs/wsj0-mix/common/src/driver.py), served by speech_amd.driver."""
import _ctn_alias  # noqa: F401
from speech_amd.driver import TrainerBase, TesterBase, EpochControl  # noqa: F401
