"""Developer probe (not product code): a simple 64-bank model of the LDS reads the intra search kernel issues per tile
row (window pair-row reads, PDPC side samples), summed over the 65 angular modes, used to pick the block / strip strides
BRS and PS of make_search_layout (intra.hip).  usage: python tools/dev/lds_bank_model.py"""
import re, sys
from collections import defaultdict
import os
src=open(os.path.join(os.path.dirname(os.path.abspath(__file__)),'..','..','uvg266_amd','csrc','intra.hip')).read()
kS=[int(x) for x in re.search(r'kSampleDisp\[32\] = \{([^}]*)\}',src).group(1).replace('\n',' ').split(',')]
kInv=[int(x) for x in re.search(r'kInvDisp\[32\] = \{([^}]*)\}',src).group(1).replace('\n',' ').split(',')]
kPre=[int(x) for x in re.search(r'kPreScale\[32\] = \{([^}]*)\}',src).group(1).replace('\n',' ').split(',')]
def cost(reqs):
    banks=defaultdict(set)
    for al in reqs:
        for a in al: banks[a%64].add(a)
    return max(len(v) for v in banks.values())
def lanes(n):
    T=8; tx_n=n//T; tiles=tx_n*tx_n
    return [(l//tiles,(l%tiles)%tx_n,(l%tiles)//tx_n) for l in range(64)]
def modes(n):
    lg=n.bit_length()-1
    out=[]
    for mode in range(2,67):
        vertical=mode>=34; md=(mode-50) if vertical else 18-mode; amd=abs(md)
        sd=(-1 if md<0 else 1)*kS[amd]; inv=kInv[amd]
        scale=min(2,lg-kPre[amd])
        pd = 2 if (md>0 and scale>=0) else 0
        out.append((sd,inv,scale,pd))
    return out
def eval_size(n,BRS,PS,wave=3):
    L=lanes(n); bpg=64//((n//8)**2); RS=2*n+4
    tot_win=tot_side=0
    for sd,inv,scale,pd in modes(n):
        for r in range(8):
            if sd<0:
                base=lambda b:(wave*bpg+b)*PS+n
            else:
                base=lambda b:b*BRS
            al=[[base(b)+((sd*(8*ty+r+1))>>5)+8*tx+k for k in range(10)] for b,tx,ty in L]
            for j in range(5): tot_win+=cost([[a[2*j],a[2*j+1]] for a in al])
            if pd==2:
                lim=min(3<<scale,n)
                for i in range(8):
                    req=[]
                    for b,tx,ty in L:
                        x=8*tx+i
                        so=(((256+(x+1)*inv)>>9)+1) if x<lim else 0
                        req.append([b*BRS+RS+8*ty+so+r])
                    tot_side+=cost(req)
    return tot_win/65/8, tot_side/65/8   # avg per row
def eval_split(n,BRS,PS):
    # returns (pos-mode window cost + side cost) for BRS ; neg-mode window cost for PS (averaged over waves 0..7)
    L=lanes(n); bpg=64//((n//8)**2); RS=2*n+4
    pos=neg=side=0
    for sd,inv,scale,pd in modes(n):
        for r in range(8):
            if sd>=0 and BRS:
                al=[[b*BRS+((sd*(8*ty+r+1))>>5)+8*tx+k for k in range(10)] for b,tx,ty in L]
                for j in range(5): pos+=cost([[a[2*j],a[2*j+1]] for a in al])
                if pd==2:
                    lim=min(3<<scale,n)
                    for i in range(8):
                        req=[]
                        for b,tx,ty in L:
                            x=8*tx+i
                            so=(((256+(x+1)*inv)>>9)+1) if x<lim else 0
                            req.append([b*BRS+RS+8*ty+so+r])
                        side+=cost(req)
            if sd<0 and PS:
                for wave in (0,3,5):
                    al=[[(wave*bpg+b)*PS+n+((sd*(8*ty+r+1))>>5)+8*tx+k for k in range(10)] for b,tx,ty in L]
                    for j in range(5): neg+=cost([[a[2*j],a[2*j+1]] for a in al])/3
    return pos/8, side/8, neg/8

if __name__ == '__main__':
    for n in (16,32):
        RS=2*n+4
        res=[]
        for BRS in range(4*RS, 4*RS+65):
            p,s,_=eval_split(n,BRS,0)
            res.append((p+s,p,s,BRS))
        res.sort()
        print(n,'BRS best',res[:5],'current',[r for r in res if r[3]==4*RS+1])
        res=[]
        for PS in range(2*n+1, 2*n+1+64):
            _,_,ng=eval_split(n,0,PS)
            res.append((ng,PS))
        res.sort()
        print(n,'PS best',res[:5],'current',[r for r in res if r[1]==2*n+1])
