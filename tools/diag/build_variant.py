import sys
sys.path.insert(0,'/root/repo')
from drake_ddp_amd import build as b
import os
os.makedirs('/root/repo/drake_ddp_amd/lib/dbg', exist_ok=True)
flags = sys.argv[2:]
b.build(extra=flags, lib='/root/repo/drake_ddp_amd/lib/dbg/libmi_%s.so' % sys.argv[1], verbose=False)
print("built", sys.argv[1])
