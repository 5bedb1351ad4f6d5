"""`dataset` of the wsj0-mix recipes (egs/wsj0-mix/common/src/dataset.py), served by speech_amd.dataset."""
import _ctn_alias  # noqa: F401  (puts the repository root on sys.path)
from speech_amd.dataset import *  # noqa: F401,F403
