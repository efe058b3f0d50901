import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["HPSDF_TRACE"] = "1"
import hpsdf_loader
H = hpsdf_loader.load()
ctx = H.Context(0)
for tg, f in ((1e-5, H.Field.union3()), (1e-7, H.Field.union3())):
    for _ in range(3):
        H.create_block(ctx, H.make_config(tg), f, 1024)
