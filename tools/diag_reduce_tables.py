"""What do the countr_reduce_table launches of a finetune step sum?  Prints every table of the (B = 8, shot_num = 3, train) plan: entries,
slabs, elements, bytes read."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import models_mae_cross as mm
m = mm.__dict__["mae_vit_base_patch16"](precision="bf16").to("cuda").train()
eng = m._engine()
orig = eng._flush_list
def spy(key):
    ops, entries = eng._defer[key]
    if not eng._sizing and entries:
        tot = sum(e[2] * e[4] * 4 for e in entries)
        print("table: %2d entries, %6.1f MB read | " % (len(entries), tot / 1e6) + ", ".join("%dx%d" % (e[2], e[4]) for e in entries))
    return orig(key)
eng._flush_list = spy
eng.plan(8, 3, True)
