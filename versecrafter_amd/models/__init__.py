from .wan_transformer3d_versecrafter import VerseCrafterWanTransformer3DModel  # noqa: F401
from .wan_text_encoder import WanT5EncoderModel, convert_hf_umt5_state_dict  # noqa: F401
from .wan_vae import AutoencoderKLWan  # noqa: F401
