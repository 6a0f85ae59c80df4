from .pipeline_wan_versecrafter import WanPipelineOutput, WanVerseCrafterPipeline  # noqa: F401
