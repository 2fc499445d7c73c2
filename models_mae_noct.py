"""Drop-in for the reference's models_mae_noct.py: same module name, factories and state_dict (see countr_amd/models_mae_noct.py)."""
from countr_amd.models_mae_noct import *  # noqa: F401,F403
from countr_amd.models_mae_noct import MaskedAutoencoderViTNoCT, mae_vit_base_patch16, mae_vit_large_patch16, mae_vit_huge_patch14  # noqa: F401
