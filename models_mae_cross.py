"""Drop-in module name used by the reference CLIs (`models_mae_cross.__dict__[args.model](...)`,
FSC_finetune_cross.py:213): re-exports the MI355X implementation."""
from countr_amd.models_mae_cross import *  # noqa: F401,F403
from countr_amd.models_mae_cross import SupervisedMAE  # noqa: F401
