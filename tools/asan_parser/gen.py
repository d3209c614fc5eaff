import sys, os, numpy as np

rng=np.random.default_rng(int(sys.argv[1]))
out=sys.argv[2]; n=int(sys.argv[3])
tokens=[b"@",b">",b"+",b"\n",b"\r\n",b" ",b"\t",b"\n\n"]
def fq(k):
    o=[]
    for i in range(k):
        L=int(rng.integers(0,120)); s=bytes(rng.choice(np.frombuffer(b"ACGTNacgtn",np.uint8),L)); q=bytes(rng.integers(33,127,L,dtype=np.uint8))
        o.append(b"@r%d x\n"%i+s+b"\n+\n"+q+b"\n")
    return b"".join(o)
def fa(k):
    o=[]
    for i in range(k):
        L=int(rng.integers(0,400)); s=bytes(rng.choice(np.frombuffer(b"ACGTN",np.uint8),L)); o.append(b">s%d\n"%i)
        for j in range(0,L,60): o.append(s[j:j+60]+b"\n")
    return b"".join(o)
def mut(d,k):
    b=bytearray(d)
    for _ in range(k):
        if not b: break
        op=int(rng.integers(0,4)); pos=int(rng.integers(0,len(b)))
        tok=tokens[int(rng.integers(0,len(tokens)))] if rng.random()<0.7 else bytes([int(rng.integers(0,256))])
        if op==0: b[pos:pos]=tok
        elif op==1: del b[pos:pos+int(rng.integers(1,8))]
        elif op==2: b[pos:pos+len(tok)]=tok
        else: b=b[:pos]
    return bytes(b)
os.makedirs(out,exist_ok=True)
for i in range(n):
    base = fq(int(rng.integers(50,600))) if rng.random()<0.6 else fa(int(rng.integers(20,200)))
    open(os.path.join(out,"f%04d.txt"%i),"wb").write(mut(base,int(rng.integers(0,6))))

# BGZF images of FASTQ text (rk_bgzf_open / rk_bgzf_fastq_records run on them in main.cpp): intact, two files concatenated (an
# empty member in the middle), and damaged ones -- headers, lengths, payload and the end of the file
import struct, zlib
def bgzf(data, block):
    parts=[]
    for lo in range(0,len(data),block):
        c=data[lo:lo+block]; co=zlib.compressobj(1,zlib.DEFLATED,-15); body=co.compress(c)+co.flush()
        parts.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00"+struct.pack("<H",len(body)+25)+body+struct.pack("<II",zlib.crc32(c)&0xffffffff,len(c)))
    parts.append(b"\x1f\x8b\x08\x04\x00\x00\x00\x00\x00\xff\x06\x00BC\x02\x00\x1b\x00\x03\x00\x00\x00\x00\x00\x00\x00\x00\x00")
    return b"".join(parts)
for i in range(max(4,n//10)):
    t=fq(int(rng.integers(20,400))); blk=int(rng.integers(40,30000))
    img=bgzf(t,blk)
    kind=i%4
    if kind==1: img=bgzf(t[:len(t)//2-3],blk)+bgzf(t[len(t)//2-3:],blk)
    elif kind==2: img=mut(img,int(rng.integers(1,4)))
    elif kind==3:
        b=bytearray(img); pos=int(rng.integers(0,max(1,len(b)-30))); b[pos]^=int(rng.integers(1,256)); img=bytes(b)
    open(os.path.join(out,"z%04d.bgzf.txt"%i),"wb").write(img)
