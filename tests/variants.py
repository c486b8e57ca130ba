"""Which tests exercise a measured-and-lost variant: explicit marking (round 6; replaces a hook that turned failures whose
message named the experimental build into skips)."""
import pytest


def lib_is_experimental():
    """Is the library the tests load (WEBSPLAT_LIB or the product build) the EXPERIMENTAL build (ws_build_flags bit 0)?"""
    import websplat
    return bool(websplat.lib.ws_build_flags() & 1)


# The measured-and-lost variants (exp_* fields of ws_context_config: WS_DEPTH_SORT=onesweep|coop, WS_BLEND_VARIANT, WS_BLEND_DMA,
# WS_BATCH_K1, WS_FOOTPRINT=ellipse, WS_TILE_SORT=wide) are compiled only into the experimental build
# (make -C web-splat_amd experimental); their tests -- and only they -- carry @pytest.mark.experimental (whole tests, or single
# parameter sets through exp_param / env_param below) and are skipped BY THAT MARKER against the product library.  Nothing
# is skipped because of the text of an exception: a product path that raises WS_ERR_UNSUPPORTED is a failure.
VARIANT_ENV = {"WS_DEPTH_SORT": ("onesweep", "coop"), "WS_BLEND_VARIANT": None, "WS_BLEND_DMA": None, "WS_BATCH_K1": None,
               "WS_FOOTPRINT": ("ellipse",), "WS_TILE_SORT": ("wide",), "WS_BLEND_ASYNC": None}


def env_is_variant(env):
    for k, v in env.items():
        if k in VARIANT_ENV and (VARIANT_ENV[k] is None and v not in ("0", "") and not (k == "WS_BATCH_K1" and v == "1")
                                 or VARIANT_ENV[k] is not None and v in VARIANT_ENV[k]):
            return True
    return False


def exp_param(*values, **kw):
    return pytest.param(*values, marks=pytest.mark.experimental, **kw)


def env_param(env):
    """A parametrize value that is a dict of WS_* switches: marked experimental iff it selects a measured-and-lost variant."""
    return exp_param(env) if env_is_variant(env) else env
