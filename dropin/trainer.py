"""Top-level `trainer` module for the reference's own main.py (`from trainer import ...`, main.py:10-12):
put this directory first on PYTHONPATH and the reference driver runs on the MI355X path."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from multimodalsignal_amd.trainer import Trainer, EarlyStopping, MsigAdam  # noqa: F401
