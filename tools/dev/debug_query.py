import os, sys, numpy as np, torch
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root); sys.path.insert(0, os.path.join(root, 'tests'))
from sound_event_detection_transformer_amd.utilities.transforms import DeviceQuery
from oracle import transforms_oracle as T
rng = np.random.RandomState(3)
data = torch.from_numpy(rng.randn(1, 1, 500, 64).astype(np.float32) * 2 + 0.5)
boxes = [np.array([[0.5, 0.3], [0.2, 0.1], [0.5, 0.256]], np.float32)]
out = DeviceQuery(False)(data.cuda(), boxes).cpu().numpy()[0]
ref, codes, mm = T.query_patches(data[0].numpy(), boxes[0], False)
for k in range(3):
    mn, mx = mm[k]
    rngf = np.float32(mx - mn)
    # recover device code
    dcode = np.rint((out[k, 0] - mn) / rngf * 255).astype(np.int64)
    print('patch', k, 'rows', T.patch_rows(boxes[0][k], 500), 'code mismatches', int((dcode != codes[k]).sum()), 'float mismatches', int((out[k] != ref[k]).sum()),
          'max abs', float(np.abs(out[k] - ref[k]).max()))
    bad = np.argwhere(out[k, 0] != ref[k, 0])
    for (y, x) in bad[:3]:
        q = codes[k][y, x]
        a = np.float32(q) / np.float32(255)
        print('   ', y, x, 'code', q, 'dev', out[k, 0, y, x].hex() if hasattr(out[k,0,y,x],'hex') else float(out[k,0,y,x]).hex(), 'ref', float(ref[k, 0, y, x]).hex(),
              'q/255', float(a).hex(), 'mul', float(np.float32(a * rngf)).hex(), 'range', float(rngf).hex(), 'mn', float(mn).hex())
