"""gfxsim.blockprof — a basic-block profile of one kernel on the interpreter, for units whose listing with line tables is not the interpreted
code (a non-inlined device function gets another prologue under -gline-tables-only: k_find_blocks / header_ok_lane): executed
wave-instructions by assembly label, with the label's static size.  Read the hot labels in tools/gfxsim/_build/<unit>.s.

    python tools/gfxsim/blockprof.py [bytes=1500000]        k_find_blocks on a text member in 128 KiB chunks

Test infrastructure only."""
import sys, os, collections, re
ROOT=os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, ROOT+'/tests', ROOT+'/tools'): sys.path.insert(0,p)
from gfxsim import harness, suite
rt = harness.use(fast_probe=True)
unit='szl_kernels_inflate_par'; ksel='k_find_blocks'
mod=[m for m in rt.modules if m.name==unit+'.s'][0]
cnt=collections.Counter()
def tr(w,I): cnt[I.line]+=1
orig=rt.launch
def launch(name,*a,**k):
    rt.trace = tr if ksel in name else None
    return orig(name,*a,**k)
rt.launch=launch
import oracle_ffi as O
from sharpziplib_amd.batch import Engine
from sharpziplib_amd import corpus as C
e=Engine(); n=int(sys.argv[1]) if len(sys.argv)>1 else 1500000
data=C.generate("enwik",0xE9,0,n); m=O.deflate(data,6)
suite._knobs(SZL_INF_CHUNK_KIB=128, SZL_INF_PAR_MIN_KIB=64)
(r,used),=e.inflate([m],[data.size]); assert r.data==data.tobytes()
src=open(ROOT+'/tools/gfxsim/_build/'+unit+'.s').read().splitlines()
# group by label
lab=None; bylab=collections.Counter(); first={}
labels={}
cur='?'
for i,l in enumerate(src,1):
    mm=re.match(r'^([\.\w\$]+):',l)
    if mm: cur=mm.group(1)
    labels[i]=cur
tot=sum(cnt.values())
size=collections.Counter()
for ln,c in cnt.items(): bylab[labels[ln]]+=c; size[labels[ln]]+=1
print("total",tot, "comp bytes", len(m))
for lb,c in bylab.most_common(25): print("%6.2f%%  %-40s static %4d  execs/instr %.0f"%(100*c/tot, lb, size[lb], c/size[lb]))
