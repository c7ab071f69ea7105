import itertools
GROUPS=[list(range(0,4))+list(range(12,16))+list(range(20,28)), list(range(4,12))+list(range(16,20))+list(range(28,32)),
        list(range(32,36))+list(range(44,48))+list(range(52,60)), list(range(36,44))+list(range(48,52))+list(range(60,64))]
def cycles(addrs):  # addrs: byte address per lane (16B reads)
    tot=0
    for grp in GROUPS:
        bank={}
        for l in grp:
            a=addrs[l]
            for d in range(4):
                b=((a//4)+d)%64
                bank.setdefault(b,set()).add((a//4)+d)
        tot+=max(len(v) for v in bank.values())
    return tot
def sim(PW, HST_bytes, swz=None, rows=40, cols=10):
    total=0; n=0
    for T in range(25):
        for c in range(5):
            addrs=[]
            for lane in range(64):
                m=lane&15; g=lane>>4
                p=16*T+m
                tap=min(2*c+(g>>1),8)
                r=p//cols+tap//3; cc=p%cols+tap%3
                pos=r*PW+cc
                a=pos*HST_bytes+16*(g&1)
                if swz: a=swz(pos,g&1,HST_bytes)
                addrs.append(a)
            total+=cycles(addrs); n+=1
    return total/n
for PW in (12,13,14,16,18,20):
    print("PW",PW,"stride 32B: avg LDS cycles per ds_read_b128 =", round(sim(PW,32),2), "(4 = conflict-free)")
# swizzle: xor the 16B half index with a bit of the position
for PW in (12,16):
    print("PW",PW,"swizzle half^=(pos>>2)&1:", round(sim(PW,32,lambda pos,h,S: pos*S+16*(h^((pos>>2)&1))),2))
    print("PW",PW,"swizzle half^=(pos>>3)&1:", round(sim(PW,32,lambda pos,h,S: pos*S+16*(h^((pos>>3)&1))),2))
    print("PW",PW,"swizzle half^=pos&1:", round(sim(PW,32,lambda pos,h,S: pos*S+16*(h^(pos&1))),2))
print("---- interleaved pitch")
for PW in (24,25,26,27,28,34):
    print("PW",PW, round(sim(PW,32),2))
# writes: ds_write_b64: assume processed 32 lanes per pass, bank conflicts among 32 lanes
def wcycles(addrs):
    tot=0
    for half in (range(0,32),range(32,64)):
        bank={}
        for l in half:
            for d in range(2):
                w=addrs[l]//4+d
                bank.setdefault(w%64,set()).add(w)
        tot+=max(len(v) for v in bank.values())
    return tot
def wsim(PW):
    t=0
    for T in range(25):
        addrs=[]
        for lane in range(64):
            pn=lane&15; g=lane>>4; p=16*T+pn
            addrs.append(((p//10+1)*PW+p%10+1)*32+8*g)
        t+=wcycles(addrs)
    return t/25
for PW in (12,26): print("write PW",PW,wsim(PW),"(2 = conflict-free)")
print("---- wide split tower: oct-planes of [rows x pitch positions][8 halves], lane (m, g) reads plane g")
def sim_wide(PW, plane_bytes):
    total=0; n=0
    for T in range(25):
        for tap in range(9):
            addrs=[]
            for lane in range(64):
                m=lane&15; g=lane>>4
                p=16*T+m
                pos=(p//10+tap//3)*PW+p%10+tap%3
                addrs.append(g*plane_bytes+pos*16)
            total+=cycles(addrs); n+=1
    return total/n
for PW,pb in ((12,8192),(26,42*26*16),(26,42*26*16+64),(26,42*26*16+128)):
    print("pitch",PW,"plane bytes",pb, round(sim_wide(PW,pb),2))
for PW,pb in ((26,17664),(12,8192+0),(18,42*18*16),(18,12288)):
    print("pitch",PW,"plane bytes",pb, round(sim_wide(PW,pb),2))
