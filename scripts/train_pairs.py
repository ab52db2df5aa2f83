"""Which (data-gradient GEMM, weight-gradient pass) shape pairs the size rules of csrc/train_mlp.hip produce at the training
levels of the four reference networks: the table of launch_pair (tl_pair_kernel instantiations). A Python restatement of
gemm_shape / wgrad_shape / top_stored / top_s_shape's size rule; `PN2_PAIR_TRACE` lab builds print what a run asks for
that the table lacks (scripts/build_mlp_labs.sh train_mlp pairtrace:-DPN2_PAIR_TRACE)."""
def tiles(c): return (c+31)//32
def gemm_shape(rows,K,N):
    tk,tn=tiles(K),tiles(N)
    ns=4 if tn>=3 else 2 if tn==2 else 1
    slabs=-(-tn//ns)
    rounds=(rows//32+7)//8
    while ns>1 and rounds*slabs<128:
        ns//=2; slabs=-(-tn//ns)
    return ns,slabs,min(max(256//slabs,1),rounds)
def wgrad_shape(rows,KI,NO,gather=False,cus=256):
    tu,tt=tiles(KI),tiles(NO)
    tus=min(tu,4); tts=min(tt,8)
    if gather and 4<tu<=6: tus=tu; tts=min(tt,2)
    while (-(-tu//tus))*(-(-tt//tts))*((rows//32+3)//4)<64 and (tus>1 or tts>1):
        if tts>=tus and tts>1: tts=(tts+1)//2
        else: tus=(tus+1)//2
    usl,tsl=-(-tu//tus),-(-tt//tts)
    nout=tus*tts; per=(nout+7)//8
    tpw=1 if per<=1 else 2 if per<=2 else 4
    upw=(2*(tus+tts)+7)//8
    slabs=usl*tsl
    blocks=rows//32
    gx=max(cus//slabs,1)
    gx=min(gx,(blocks+3)//4)
    return tpw,upw,tus,tts,slabs,gx
# levels: name, rows, widths(cin..), pooled(ns or 0), grouped(bool), cfeat
L=[]
def sa(name,b,m,ns,c,w): L.append((name,b*m*ns,[3+c]+w,ns,True,c))
def ga(name,b,n,c,w): L.append((name,b*n,[3+c]+w,n,True,c))
def fp(name,b,n,cin,w): L.append((name,b*n,[cin]+w,0,False,0))
sa('sem SA1',8,1024,32,0,[32,32,64]); sa('sem SA2',8,256,32,64,[64,64,128]); sa('sem SA3',8,64,32,128,[128,128,256]); sa('sem SA4',8,16,32,256,[256,256,512])
fp('sem FP1',8,64,768,[256,256]); fp('sem FP2',8,256,384,[256,256]); fp('sem FP3',8,1024,320,[256,128]); fp('sem FP4',8,8192,128,[128,128,128])
sa('part SA1',16,512,64,3,[64,64,128]); sa('part SA2',16,128,64,128,[128,128,256]); ga('part SA3',16,128,256,[256,512,1024])
fp('part FP1',16,128,1280,[256,256]); fp('part FP2',16,512,384,[256,128]); fp('part FP3',16,2048,134,[128,128])
sa('cls SA1',32,512,32,0,[64,64,128]); sa('cls SA2',32,128,64,128,[128,128,256]); ga('cls SA3',32,128,256,[256,512,1024])
pairs={}
for name,rows,w,pool,grouped,c in L:
    nl=len(w)-1
    stored = not pool or rows*w[-1]*4 < (32<<20)
    print(name,rows,w,'stored' if stored else 'ztop')
    for l in range(nl-1,-1,-1):
        cin,cout=w[l],w[l+1]
        top = pool and l==nl-1
        if top and not stored:
            K=cin; NF=cout; ld=tiles(NF)*32+tiles(K)*32+32
            sparse = rows*K >= (1<<24) and pool*K <= 16384         # routed part on the vector units (top_s_shape's size rule)
            g=gemm_shape(rows,tiles(NF)*32+K,K); wg=wgrad_shape(rows,K,(0 if sparse else tiles(NF)*32)+tiles(K)*32+32)
            key=('FILL',g[0],'TOP',wg[0],wg[1]); 
        elif l==0 and grouped:
            print('   l0 grouped (per-point / coords paths)'); continue
        else:
            g=gemm_shape(rows,cout,cin); wg=wgrad_shape(rows,cin,cout)
            key=('DZPOOL' if top else 'DZ',g[0],'DZPOOL' if top else 'DZ',wg[0],wg[1])
        pairs.setdefault(key,[]).append(name+' l%d'%l)
        print('   l%d %s gemm ns%d slabs%d gx%d | wgrad tpw%d upw%d tus%d tts%d slabs%d gx%d'%(l,key[0],g[0],g[1],g[2],wg[0],wg[1],wg[2],wg[3],wg[4],wg[5]))
print()
for k,v in sorted(pairs.items()): print(k,len(v),v)
