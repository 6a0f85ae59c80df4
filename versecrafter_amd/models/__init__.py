from .wan_transformer3d_versecrafter import VerseCrafterWanTransformer3DModel  # noqa: F401
