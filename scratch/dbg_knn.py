import sys, importlib, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'oracle')
import oracle as O
synth = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.synth")
pl = importlib.import_module("lk-s-2022-estimacija-pokreta_amd.pipeline")
H,W,ch,cw = 40,48,5,6
img1,img2,_ = synth.make_pair(H,W,seed=H*W,amp_x=0.1*W,amp_y=0.1*H)
df = pl.DiscreteFlow(H,W,ch,cw,seed=1234+H)
p = O.make_params(H,W,ch,cw,seed=1234+H)
df.load_pair(img1,img2)
d1,d2 = O.daisy(img1),O.daisy(img2)
df.generisi()
pr,lc,npr,bl = O.knn_proposals(p,d1,d2)
st = df.host_state()
diff = np.argwhere((st['proposals']!=pr).any(-1))
print('ndiff', len(diff), 'of', (npr.sum()))
for y,x,l in diff[:10]:
    print(y,x,l,'gpu',st['proposals'][y,x,l//5*5:l//5*5+5].tolist(),'orc',pr[y,x,l//5*5:l//5*5+5].tolist(), 'lc', st['lcosts'][y,x,l//5*5:l//5*5+5], lc[y,x,l//5*5:l//5*5+5])
y,x,l = diff[0]
# recompute dist for that pixel/cell
ncy = H//ch
cimin=max(0,x//cw-2); cjmin=max(0,y//ch-2); cjmax=min(ncy-1,y//ch+2)
g = l//5; ci = cimin + g//(cjmax-cjmin+1); cj = cjmin + g%(cjmax-cjmin+1)
idx,dist = O.knn_cell(p,d1[y,x],d2,ci,cj); print('cell',ci,cj,'oracle idx',idx,dist)
pts = d2[cj*ch:(cj+1)*ch, ci*cw:(ci+1)*cw].reshape(-1,68)
dd = ((pts.astype(np.float64)-d1[y,x].astype(np.float64))**2).sum(1); o=np.argsort(dd); print(o[:8], dd[o[:8]])
