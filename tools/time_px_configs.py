"""(round 5) The KPM-preconditioned batch iteration p/x-fused vs unfused, one stream vs two, for configs D (and C, E: edit the tuple): us per iteration, slices per wave, fused flag.  usage: python3 tools/time_px_configs.py"""
import os, sys, json
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__))))
import validate_form_choice as v
import sys as _s
for tag in (_s.argv[1:] or ["D"]):
    for n in (64, 128, 256, 512):
        row = {}
        for name, env, w in (("px 1 stream", {"ELPH_SPLIT_STREAMS": "0"}, "prec"), ("px 2 streams", {}, "prec2"),
                             ("unfused 1 stream", {"ELPH_FUSE_PX": "0", "ELPH_SPLIT_STREAMS": "0"}, "prec"),
                             ("px T=16", {"ELPH_CHUNK_T": "16", "ELPH_SPLIT_STREAMS": "0"}, "prec"),
                             ("unfused T=16", {"ELPH_CHUNK_T": "16", "ELPH_FUSE_PX": "0", "ELPH_SPLIT_STREAMS": "0"}, "prec")):
            r = v.child(tag, n, w, env)
            row[name] = (r.get("us") and round(r["us"], 1), r.get("T"), r.get("px"))
        print(tag, n, row, flush=True)
