"""versecrafter_amd: MI355X-native (gfx950) engine for the VerseCrafter denoise step.

Only the hot path of ztitomir/VerseCrafter lives here: the Wan2.1 DiT + GeoAdapter forward behind the
reference's own Python interface (`models.VerseCrafterWanTransformer3DModel`,
`pipeline.WanVerseCrafterPipeline`), computed by hand-written HIP kernels in `libvcengine.so`
(`csrc/`, C ABI in `include/vcengine.h`).  There is no CPU implementation in this package.
"""
__all__ = ["models", "pipeline", "ops", "dist", "utils"]
