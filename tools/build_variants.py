"""Dev tool: build experiment variants of the library for A/B runs (tools/ab.sh with --lib-variant NAME).
usage: python tools/build_variants.py name1=-DFOO,-DBAR name2=-DBAZ ..."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from triplaneturbo_amd import _lib  # noqa: E402

import json  # noqa: E402

# TT_SRC_FLAGS='{"name": {"tt_backward.hip": ["-mllvm", "-amdgpu-sched-strategy=iterative-ilp"]}}' overrides the
# per-translation-unit flags of variant "name"
SRC = json.loads(os.environ.get("TT_SRC_FLAGS", "{}"))
for spec in sys.argv[1:]:
    name, _, defs = spec.partition("=")
    flags = [d for d in defs.split(",") if d]
    # "@plain" drops the per-translation-unit flags of _lib.SOURCE_FLAGS (A/B of those flags themselves)
    print(_lib.build(force=True, variant=name, tuning=bool(os.environ.get("TT_VARIANT_TUNING")), defines=[f for f in flags if f != "@plain"],
                     source_flags=SRC.get(name, {} if "@plain" in flags else None)))
