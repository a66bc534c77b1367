"""Builder-side probes only: engine options from the environment of the PROBE (FBR_OPT_<KEY>=value -> fbr_model_set_option(key, value) on
every Engine the probe creates).  The library itself never reads the environment; `import _opts` at the top of a probe keeps the A/B
scripts (tools/ab_run.sh ...) a one-liner per variant."""
import os

from flobaroid_amd import _lib

for _k, _v in os.environ.items():
    if _k.startswith("FBR_OPT_"):
        _lib.DEFAULT_OPTIONS[_k[len("FBR_OPT_"):].lower()] = float(_v)
if _lib.DEFAULT_OPTIONS:
    print("[tools/_opts] engine options:", _lib.DEFAULT_OPTIONS, flush=True)
