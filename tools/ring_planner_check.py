"""The 16-bit ring kernel's launch planner against a shape sweep (profiles/r05_ring_shape_sweep_7_batch_shapes.txt, made with
tools/shape_sweep.py 4,...,11 bf16x3 at seven batch shapes): a linear refit of the tile cost, how close the shipped model's picks are
to the measured best shape, and the most a head + tail split could gain with PERFECT knowledge of every tile time.
    python tools/ring_planner_check.py [sweep file]"""
import sys
import re, numpy as np, math
CIN={4:67,5:100,6:150,7:225,8:337,9:505,10:757,11:1135}; COUT={4:100,5:150,6:225,7:337,8:505,9:757,10:1135,11:1702}
NUM_CU=256; CLK=2100.0   # cycles per us (ring kernels run ~1.9-2.2 GHz; a scale only)
def pitch(L,layer):
    fine=math.ceil(L/1024)*1024; coarse=math.ceil(L/4096)*4096
    return (coarse if layer>=9 else fine)>>layer
rows=[]; B=L=None
for line in open(sys.argv[1] if len(sys.argv) > 1 else __import__('os').path.join(__import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))), 'profiles', 'r05_ring_shape_sweep_7_batch_shapes.txt')):
    m=re.match(r'== B=(\d+) L=(\d+)',line)
    if m: B,L=int(m.group(1)),int(m.group(2)); continue
    m=re.match(r'\((\d), (\d), (\d), (\d)\)\s+\S+\s+(.*)',line)
    if not m: continue
    wm,wn,mt,nt=[int(m.group(i)) for i in range(1,5)]
    for lm in re.finditer(r'L(\d+)=([\d.]+)',m.group(5)):
        l=int(lm.group(1)); t=float(lm.group(2))*1e3
        npan=math.ceil(CIN[l]/32); n16=math.ceil(COUT[l]/16)
        r=B*pitch(L,l); bm=wm*16*mt; bnt=wn*nt
        tiles=math.ceil(r/bm)*math.ceil(n16/bnt)
        rows.append(dict(B=B,L=L,l=l,wm=wm,wn=wn,mt=mt,nt=nt,t=t,npan=npan,tiles=tiles,bm=bm,bn=bnt*16))
def design(r):
    rounds=math.ceil(r['tiles']/NUM_CU); ns=3*r['npan']*rounds
    mf=3*r['mt']*r['nt']*16*2
    dma=((r['bm']+8)/3+r['bn'])*128
    fill=min(1.0,r['tiles']/NUM_CU)
    return rounds, ns*mf, [1.0, ns, ns*dma, ns*dma*fill, ns*(r['mt']+r['nt']), rounds, rounds*r['mt']*r['nt'], rounds*r['bm']*r['bn']/1024.0]
X=[];Y=[];M=[]
for r in rows:
    rounds,mf,f=design(r); X.append(f); Y.append(r['t']*CLK-mf); M.append((r,mf))
X=np.array(X);Y=np.array(Y)
coef,*_=np.linalg.lstsq(X,Y,rcond=None)
names=["launch","per sub-stage","dma byte","dma byte*fill","per sub*(mt+nt)","per tile","per tile*mt*nt","per tile*KB out"]
for n,c in zip(names,coef): print("  %-18s %12.4f"%(n,c))
T=np.array([m[0]['t']*CLK for m in M]); P=X@coef+np.array([m[1] for m in M]); e=P/T
print("n=%d ratio mean %.3f std %.3f min %.3f max %.3f"%(len(e),e.mean(),e.std(),e.min(),e.max()))
bad=[(m[0],ee) for m,ee in zip(M,e) if abs(ee-1)>0.15]
print(len(bad),"off by >15%")
for r,ee in bad[:25]: print("   B",r['B'],"L",r['L'],"layer",r['l'],(r['wm'],r['wn'],r['mt'],r['nt']),"t %.0f tiles %d ratio %.2f"%(r['t'],r['tiles'],ee))
# how often does the fitted model pick within 3% of the best shape per (B,L,layer)?
from collections import defaultdict
g=defaultdict(list)
for (r,mf),p in zip(M,P): g[(r['B'],r['L'],r['l'])].append((r['t'],p,(r['wm'],r['wn'],r['mt'],r['nt'])))
loss=[]
for k,v in g.items():
    best=min(v)[0]; pick=min(v,key=lambda x:x[1]); loss.append(pick[0]/best)
print("picks: mean loss %.3f, worst %.3f, within 3%%: %d of %d"%(np.mean(loss),max(loss),sum(1 for x in loss if x<=1.03),len(loss)))
print("---- current planner vs best, and the head+tail upper bound with measured tile times")
def cur_tile_cost(r):
    mt,nt,bm,bnt=r['mt'],r['nt'],r['bm'],r['bn']//16
    mfma=3.0*mt*nt*16*2; dma=((bm+8)/3.0+bnt*16.0)*128.0/24.0; ldsr=2.0*8.0*(mt+nt)*1024.0/256.0*1.2
    sub=max(mfma,dma,ldsr)+350.0
    return 3.0*r['npan']*sub+1500.0+60.0*mt*nt*1.5
loss=[];
LAUNCH=7.0
tile_time=defaultdict(list)
for r in rows:
    rounds=math.ceil(r['tiles']/NUM_CU)
    tile_time[(r['l'],r['wm'],r['wn'],r['mt'],r['nt'])].append((r['t']-LAUNCH)/rounds)
tt={k:float(np.median(v)) for k,v in tile_time.items()}
gains=[]
for k,v in g.items():
    B_,L_,l=k
    rs=[r for r in rows if (r['B'],r['L'],r['l'])==k]
    best=min(r['t'] for r in rs)
    pick=min(rs,key=lambda r: math.ceil(r['tiles']/NUM_CU)*cur_tile_cost(r))
    loss.append(pick['t']/best)
    # head + tail with measured tile times
    r0=B_*pitch(L_,l); n16=math.ceil(COUT[l]/16)
    single=min(LAUNCH+math.ceil(math.ceil(r0/r['bm'])*math.ceil(n16/(r['bn']//16))/NUM_CU)*tt[(l,r['wm'],r['wn'],r['mt'],r['nt'])] for r in rs)
    bestsplit=single
    for h in rs:
        nn=math.ceil(n16/(h['bn']//16)); nm=math.ceil(r0/h['bm']); tiles=nm*nn; full=tiles//NUM_CU
        if full<1 or tiles%NUM_CU==0: continue
        m1=full*NUM_CU//nn
        if m1<1 or m1>=nm: continue
        head_rounds=math.ceil(m1*nn/NUM_CU)
        rem=r0-m1*h['bm']
        for t_ in rs:
            tiles_t=math.ceil(rem/t_['bm'])*math.ceil(n16/(t_['bn']//16))
            c=LAUNCH+head_rounds*tt[(l,h['wm'],h['wn'],h['mt'],h['nt'])]+3.0+math.ceil(tiles_t/NUM_CU)*tt[(l,t_['wm'],t_['wn'],t_['mt'],t_['nt'])]
            bestsplit=min(bestsplit,c)
    gains.append((k,single,bestsplit))
print("current planner picks: mean loss %.3f worst %.3f within 3%%: %d of %d"%(np.mean(loss),max(loss),sum(1 for x in loss if x<=1.03),len(loss)))
byshape=defaultdict(lambda:[0.0,0.0])
for (B_,L_,l),s_,b_ in gains:
    byshape[(B_,L_)][0]+=s_; byshape[(B_,L_)][1]+=b_
for k,(s_,b_) in sorted(byshape.items()): print("  B=%d L=%d: layers 4-11 single %.0f us, best head+tail %.0f us (%.1f %%)"%(k[0],k[1],s_,b_,100*(b_/s_-1)))
