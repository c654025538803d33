"""CPU oracle (test infrastructure, never on the product path): the JPEG file Canvas::to_jpeg writes
(rendering/canvas.rs:92-114 -> `image` 0.24.6, ImageOutputFormat::Jpeg(100); Cargo.lock:234-235).

The `image` crate's source is not in the reference checkout; this is a restatement of its baseline encoder (JFIF 1.02
header, SOF0 with three 1x1 components, all-ones quantisation tables at quality 100, the standard Huffman tables,
f32 RGB -> YCbCr with the JFIF coefficients scaled for 255, libjpeg-9-style integer DCT scaled by 8, truncating
division, EOB only when the last coefficient is zero, padding with ones).  PINNED: the reference ships the PPM and the
JPEG of five renderings; `jpeg_bytes(pixels of the PPM)` equals each of those JPEG files byte for byte
(tests/test_jpeg.py) -- which also settled the one detail no description gives: the rounding term of the odd part in
the column pass is the row pass's (1 << 10), not libjpeg's (1 << 14)."""
import struct

import numpy as np

STD_LUMA_DC_LEN=[0,1,5,1,1,1,1,1,1,0,0,0,0,0,0,0]; STD_LUMA_DC_VAL=list(range(12))
STD_CHROMA_DC_LEN=[0,3,1,1,1,1,1,1,1,1,1,0,0,0,0,0]; STD_CHROMA_DC_VAL=list(range(12))
STD_LUMA_AC_LEN=[0,2,1,3,3,2,4,3,5,5,4,4,0,0,1,0x7d]
STD_LUMA_AC_VAL=[0x01,0x02,0x03,0x00,0x04,0x11,0x05,0x12,0x21,0x31,0x41,0x06,0x13,0x51,0x61,0x07,0x22,0x71,0x14,0x32,0x81,0x91,0xa1,0x08,0x23,0x42,0xb1,0xc1,0x15,0x52,0xd1,0xf0,0x24,0x33,0x62,0x72,0x82,0x09,0x0a,0x16,0x17,0x18,0x19,0x1a,0x25,0x26,0x27,0x28,0x29,0x2a,0x34,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,0x49,0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x83,0x84,0x85,0x86,0x87,0x88,0x89,0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,0xc4,0xc5,0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,0xe1,0xe2,0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf1,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,0xf9,0xfa]
STD_CHROMA_AC_LEN=[0,2,1,2,4,4,3,4,7,5,4,4,0,1,2,0x77]
STD_CHROMA_AC_VAL=[0x00,0x01,0x02,0x03,0x11,0x04,0x05,0x21,0x31,0x06,0x12,0x41,0x51,0x07,0x61,0x71,0x13,0x22,0x32,0x81,0x08,0x14,0x42,0x91,0xa1,0xb1,0xc1,0x09,0x23,0x33,0x52,0xf0,0x15,0x62,0x72,0xd1,0x0a,0x16,0x24,0x34,0xe1,0x25,0xf1,0x17,0x18,0x19,0x1a,0x26,0x27,0x28,0x29,0x2a,0x35,0x36,0x37,0x38,0x39,0x3a,0x43,0x44,0x45,0x46,0x47,0x48,0x49,0x4a,0x53,0x54,0x55,0x56,0x57,0x58,0x59,0x5a,0x63,0x64,0x65,0x66,0x67,0x68,0x69,0x6a,0x73,0x74,0x75,0x76,0x77,0x78,0x79,0x7a,0x82,0x83,0x84,0x85,0x86,0x87,0x88,0x89,0x8a,0x92,0x93,0x94,0x95,0x96,0x97,0x98,0x99,0x9a,0xa2,0xa3,0xa4,0xa5,0xa6,0xa7,0xa8,0xa9,0xaa,0xb2,0xb3,0xb4,0xb5,0xb6,0xb7,0xb8,0xb9,0xba,0xc2,0xc3,0xc4,0xc5,0xc6,0xc7,0xc8,0xc9,0xca,0xd2,0xd3,0xd4,0xd5,0xd6,0xd7,0xd8,0xd9,0xda,0xe2,0xe3,0xe4,0xe5,0xe6,0xe7,0xe8,0xe9,0xea,0xf2,0xf3,0xf4,0xf5,0xf6,0xf7,0xf8,0xf9,0xfa]
ZIGZAG=[0,1,8,16,9,2,3,10,17,24,32,25,18,11,4,5,12,19,26,33,40,48,41,34,27,20,13,6,7,14,21,28,35,42,49,56,57,50,43,36,29,22,15,23,30,37,44,51,58,59,52,45,38,31,39,46,53,60,61,54,47,55,62,63]
def build(lens, vals):
    codes={}; code=0; k=0
    for l in range(1,17):
        for _ in range(lens[l-1]):
            codes[vals[k]]=(code,l); code+=1; k+=1
        code<<=1
    return codes
C=13;P=2
F=dict(f0298=2446,f0390=3196,f0541=4433,f0765=6270,f0899=7373,f1175=9633,f1501=12299,f1847=15137,f1961=16069,f2053=16819,f2562=20995,f3072=25172)
def fdct(s, oddfudge2):
    # s: (N,64) int; vectorised over blocks
    c=np.zeros_like(s, dtype=np.int64); s=s.astype(np.int64)
    for y in range(8):
        r=s[:,y*8:y*8+8]
        t0=r[:,0]+r[:,7];t1=r[:,1]+r[:,6];t2=r[:,2]+r[:,5];t3=r[:,3]+r[:,4]
        t10=t0+t3;t12=t0-t3;t11=t1+t2;t13=t1-t2
        t0=r[:,0]-r[:,7];t1=r[:,1]-r[:,6];t2=r[:,2]-r[:,5];t3=r[:,3]-r[:,4]
        c[:,y*8]=(t10+t11-8*128)<<P; c[:,y*8+4]=(t10-t11)<<P
        z1=(t12+t13)*F['f0541']+(1<<(C-P-1))
        c[:,y*8+2]=(z1+t12*F['f0765'])>>(C-P); c[:,y*8+6]=(z1-t13*F['f1847'])>>(C-P)
        t12=t0+t2;t13=t1+t3
        z1=(t12+t13)*F['f1175']+(1<<(C-P-1))
        t12=t12*(-F['f0390'])+z1; t13=t13*(-F['f1961'])+z1
        z1=(t0+t3)*(-F['f0899']); a0=t0*F['f1501']+z1+t12; a3=t3*F['f0298']+z1+t13
        z1=(t1+t2)*(-F['f2562']); a1=t1*F['f3072']+z1+t13; a2=t2*F['f2053']+z1+t12
        c[:,y*8+1]=a0>>(C-P);c[:,y*8+3]=a1>>(C-P);c[:,y*8+5]=a2>>(C-P);c[:,y*8+7]=a3>>(C-P)
    for x in range(7,-1,-1):
        g=lambda k:c[:,x+8*k].copy()
        t0=g(0)+g(7);t1=g(1)+g(6);t2=g(2)+g(5);t3=g(3)+g(4)
        t10=t0+t3+(1<<(P-1));t12=t0-t3;t11=t1+t2;t13=t1-t2
        t0=g(0)-g(7);t1=g(1)-g(6);t2=g(2)-g(5);t3=g(3)-g(4)
        c[:,x]=(t10+t11)>>P; c[:,x+32]=(t10-t11)>>P
        z1=(t12+t13)*F['f0541']+(1<<(C+P-1))
        c[:,x+16]=(z1+t12*F['f0765'])>>(C+P); c[:,x+48]=(z1-t13*F['f1847'])>>(C+P)
        t12=t0+t2;t13=t1+t3
        z1=(t12+t13)*F['f1175']+(1<<((C+P-1) if oddfudge2 else (C-P-1)))
        t12=t12*(-F['f0390'])+z1; t13=t13*(-F['f1961'])+z1
        z1=(t0+t3)*(-F['f0899']); a0=t0*F['f1501']+z1+t12; a3=t3*F['f0298']+z1+t13
        z1=(t1+t2)*(-F['f2562']); a1=t1*F['f3072']+z1+t13; a2=t2*F['f2053']+z1+t12
        c[:,x+8]=a0>>(C+P);c[:,x+24]=a1>>(C+P);c[:,x+40]=a2>>(C+P);c[:,x+56]=a3>>(C+P)
    return c
def ycbcr(px, variant):
    r=px[...,0].astype(np.float32);g=px[...,1].astype(np.float32);b=px[...,2].astype(np.float32)
    f=np.float32
    if variant==0:
        m=f(255.0)
        y=f(76.245)/m*r+f(149.685)/m*g+f(29.07)/m*b
        cb=f(-43.0185)/m*r-f(84.4815)/m*g+f(127.5)/m*b+f(128.)
        cr=f(127.5)/m*r-f(106.7685)/m*g-f(20.7315)/m*b+f(128.)
    else:
        y=f(0.299)*r+f(0.587)*g+f(0.114)*b
        cb=f(-0.1687)*r-f(0.3313)*g+f(0.5)*b+f(128.)
        cr=f(0.5)*r-f(0.4187)*g-f(0.0813)*b+f(128.)
    cl=lambda v:np.clip(np.trunc(v),0,255).astype(np.int64)
    return cl(y),cl(cb),cl(cr)
def encode(px, variant=0, oddfudge2=False):
    h,w,_=px.shape
    H8=(h+7)//8*8; W8=(w+7)//8*8
    ys=np.minimum(np.arange(H8),h-1); xs=np.minimum(np.arange(W8),w-1)
    p=px[ys][:,xs]
    comps=ycbcr(p,variant)
    blocks=[]
    for c in comps:
        bl=c.reshape(H8//8,8,W8//8,8).transpose(0,2,1,3).reshape(-1,64)
        d=fdct(bl,oddfudge2)
        q=np.trunc(d/8.0).astype(np.int64)  # integer division truncating toward zero
        blocks.append(q)
    tabs=[(build(STD_LUMA_DC_LEN,STD_LUMA_DC_VAL),build(STD_LUMA_AC_LEN,STD_LUMA_AC_VAL)),(build(STD_CHROMA_DC_LEN,STD_CHROMA_DC_VAL),build(STD_CHROMA_AC_LEN,STD_CHROMA_AC_VAL))]
    out=bytearray(); acc=0; nb=0
    def put(v,n):
        nonlocal acc,nb
        if n==0: return
        acc=(acc<<n)|(v&((1<<n)-1)); nb+=n
        while nb>=8:
            byte=(acc>>(nb-8))&0xff; out.append(byte)
            if byte==0xff: out.append(0)
            nb-=8
        acc&=(1<<nb)-1
    def coef(v):
        a=abs(int(v)); size=a.bit_length()
        val=int(v) if v>=0 else int(v)-1
        return size,val&((1<<size)-1)
    prev=[0,0,0]
    n=blocks[0].shape[0]
    for i in range(n):
        for ci in range(3):
            blk=blocks[ci][i]; dc,ac=tabs[0 if ci==0 else 1]
            diff=int(blk[0])-prev[ci]; prev[ci]=int(blk[0])
            s,v=coef(diff); code,l=dc[s]; put(code,l); put(v,s)
            run=0
            for k in ZIGZAG[1:]:
                if blk[k]==0: run+=1
                else:
                    while run>15:
                        code,l=ac[0xf0]; put(code,l); run-=16
                    s,v=coef(blk[k]); code,l=ac[(run<<4)|s]; put(code,l); put(v,s); run=0
            if blk[63]==0:
                code,l=ac[0]; put(code,l)
    put(0x7f,7) if nb else None
    return bytes(out)


def _segment(marker: int, body: bytes) -> bytes:
    return bytes([0xff, marker]) + struct.pack(">H", len(body) + 2) + body


def _dht(klass: int, dest: int, lens, vals) -> bytes:
    return _segment(0xc4, bytes([(klass << 4) | dest]) + bytes(lens) + bytes(vals[:sum(lens)]))


def jpeg_bytes(pixels: np.ndarray) -> bytes:
    """The whole file for an [h, w, 3] uint8 image (Canvas::to_rgb_image's output)."""
    h, w, _ = pixels.shape
    out = b"\xff\xd8"
    out += _segment(0xe0, b"JFIF\0" + bytes([1, 2, 0, 0, 1, 0, 1, 0, 0]))
    out += _segment(0xc0, bytes([8]) + struct.pack(">HH", h, w) + bytes([3, 1, 0x11, 0, 2, 0x11, 1, 3, 0x11, 1]))
    out += _segment(0xdb, bytes([0]) + bytes([1] * 64))
    out += _segment(0xdb, bytes([1]) + bytes([1] * 64))
    out += _dht(0, 0, STD_LUMA_DC_LEN, STD_LUMA_DC_VAL) + _dht(1, 0, STD_LUMA_AC_LEN, STD_LUMA_AC_VAL)
    out += _dht(0, 1, STD_CHROMA_DC_LEN, STD_CHROMA_DC_VAL) + _dht(1, 1, STD_CHROMA_AC_LEN, STD_CHROMA_AC_VAL)
    out += _segment(0xda, bytes([3, 1, 0x00, 2, 0x11, 3, 0x11, 0, 0x3f, 0]))
    return out + encode(pixels) + b"\xff\xd9"
