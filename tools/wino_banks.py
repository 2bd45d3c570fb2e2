"""Search of LDS layouts for K17's 32-channel raw chunk: which (pixel pitch, row pitch, column split, XOR swizzle) makes the transform's
ds_read_b128 (16-lane groups {0-3,12-15,20-27}, {4-11,16-19,28-31}; lanes = tiles two pixels apart) conflict free.  Prints the best:
(worst ways, mean ways, split, pitch in 16-byte slots, row pitch in pixels, row XOR, column XOR)."""
import itertools
GROUPS=[[0,1,2,3,12,13,14,15,20,21,22,23,24,25,26,27],[4,5,6,7,8,9,10,11,16,17,18,19,28,29,30,31]]
def conflicts(slotfn):
    worst=0; tot=0; n=0
    for tb in (0,1):
      for r in (0,1,2,3):
        for c in range(4):
          for q in range(8):     # quad = 4s+2h+hf fixed per instruction/half-wave
            for g in GROUPS:
                banks={}
                for lane in g:
                    tyl, tx = lane>>3, lane&7
                    prow=2*(4*tb+tyl)+r; pcol=2*tx+c
                    s16=slotfn(prow,pcol,q)
                    banks.setdefault(s16%16,set()).add(s16)
                w=max(len(v) for v in banks.values())
                worst=max(worst,w); tot+=w; n+=1
    return worst, tot/n
best=[]
for split in (0,1):
  for P16 in (8,9):
    for Pr in (18,19,20):
      for xr in range(0,4):      # xor source: 0 none, 1: (prow>>1)&3, 2: (prow>>1)&7, 3: prow&7
        for xc in range(0,3):    # extra xor from column: 0 none, 1: ((pcol>>2)&1)<<2, 2: ((pcol>>1)&... 
          def mk(split=split,P16=P16,Pr=Pr,xr=xr,xc=xc):
            def f(prow,pcol,q):
                idx = prow*Pr + ((pcol&1)*9 + (pcol>>1) if split else pcol)
                sg = [0,(prow>>1)&3,(prow>>1)&7,prow&7][xr]
                sc = [0,((pcol>>2)&1)<<2, ((pcol>>2)&3)<<1][xc]
                return idx*P16 + ((q ^ sg ^ sc) & 7)
            return f
          w,a=conflicts(mk())
          best.append((w,a,split,P16,Pr,xr,xc))
best.sort()
for b in best[:12]: print(b)
