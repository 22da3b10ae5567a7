import sys, torch
import os; R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
from test_gpu_prepack import _trainer
from canonicalsg2im_amd import ops
tr, batches = _trainer(True)
orig_note = ops._note_pack_request
def note(w, bd, var):
    owner = ops._PACK_OWNER.get(w.data_ptr())
    print("MISS", tuple(w.shape), tuple(w.stride()), bd, var, "owner" if owner else "NO-OWNER", owner[1] if owner else "", type(owner[0]()).__name__ if owner else "")
    return orig_note(w, bd, var)
ops._note_pack_request = note
for it in range(3):
    print("=== step", it)
    tr.step(batches[it % 2])
    print("parked left:", len(ops._PREPACKED))
names = {id(m): n for n, m in tr.model.named_modules()}
for n, m in tr.model.named_modules():
    if "_pp_plan" in m.__dict__: print(n, m.__dict__["_pp_plan"])
