/* lzsim.c -- design-space explorer for the GPU deflate match finder (NOT product code).
 * Simulates: per 65280-byte block, positions processed in chunks of T (= lanes working in
 * lock-step); every position looks up W most-recent candidates in a set-associative hash
 * table that only contains positions of EARLIER chunks (+ explicit short distances), then the
 * chunk's positions are inserted.  Parse: greedy or 1-step lazy.  Cost model: entropy-coded
 * litlen/dist symbols (dynamic Huffman ~ entropy) + extra bits.
 * usage: lzsim file HASH_BITS WAYS T MINHASH(3|4) LAZY(0|1) MAXCMP SHORTD
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <zlib.h>
#define BS 65280
static int lsym(int len){ static const int base[29]={3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258}; int s=28; while(base[s]>len) s--; return s; }
static int lext(int s){ static const int e[29]={0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0}; return e[s]; }
static int dsym(int d){ static const int base[30]={1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577}; int s=29; while(base[s]>d) s--; return s; }
static int dext(int s){ return s<4?0:(s-2)>>1; }
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); int HB=atoi(argv[2]),W=atoi(argv[3]),T=atoi(argv[4]),MH=atoi(argv[5]),LAZY=atoi(argv[6]),MAXCMP=atoi(argv[7]),SHORTD=atoi(argv[8]);
  static uint8_t buf[BS+8]; size_t n; double total_bits=0, zbytes=0, inbytes=0; long nblk=0, ntok=0, nmatch=0;
  int HS=1<<HB; uint16_t*tab=malloc(sizeof(uint16_t)*HS*W); uint8_t*cnt=malloc(HS);
  int *mlen=malloc(sizeof(int)*BS), *mdist=malloc(sizeof(int)*BS);
  while((n=fread(buf,1,BS,f))>0){
    memset(buf+n,0,8); memset(tab,0xff,sizeof(uint16_t)*HS*W); memset(cnt,0,HS);
    for(size_t c0=0;c0<n;c0+=T){
      size_t c1=c0+T>n?n:c0+T;
      for(size_t p=c0;p<c1;p++){
        int best=0,bd=0; size_t maxl=n-p>258?258:n-p;
        if(maxl>=3){
          uint32_t v; memcpy(&v,buf+p,4); uint32_t h=((MH==3?(v&0xffffff):v)*2654435761u)>>(32-HB);
          for(int w=0;w<W;w++){ uint16_t c=tab[h*W+w]; if(c==0xffff) continue; size_t l=0; size_t ml=maxl<(size_t)MAXCMP?maxl:MAXCMP; while(l<ml&&buf[c+l]==buf[p+l]) l++; int d=p-c; if((int)l>best||((int)l==best&&d<bd)){best=l;bd=d;} }
          for(int d=1;d<=SHORTD&&(size_t)d<=p;d++){ size_t l=0; size_t ml=maxl<(size_t)MAXCMP?maxl:MAXCMP; while(l<ml&&buf[p-d+l]==buf[p+l]) l++; if((int)l>best){best=l;bd=d;} }
        }
        if(best<3||(best==3&&bd>4096)) best=0;
        mlen[p]=best; mdist[p]=bd;
      }
      for(size_t p=c0;p<c1&&p+3<n;p++){ uint32_t v; memcpy(&v,buf+p,4); uint32_t h=((MH==3?(v&0xffffff):v)*2654435761u)>>(32-HB); tab[h*W+(cnt[h]++%W)]=p; }
    }
    /* parse */
    long lf[286]={0},df[30]={0}; double extra=0; size_t p=0;
    while(p<n){
      int l=mlen[p];
      if(l&&LAZY&&p+1<n&&mlen[p+1]>l){ l=0; }
      if(l){ int s=lsym(l); lf[257+s]++; extra+=lext(s); int ds=dsym(mdist[p]); df[ds]++; extra+=dext(ds); p+=l; nmatch++; }
      else { lf[buf[p]]++; p++; }
      ntok++;
    }
    lf[256]=1; double bits=extra; long lt=0,dt=0; for(int i=0;i<286;i++) lt+=lf[i]; for(int i=0;i<30;i++) dt+=df[i];
    for(int i=0;i<286;i++) if(lf[i]) bits+=lf[i]*-log2((double)lf[i]/lt);
    for(int i=0;i<30;i++) if(df[i]) bits+=df[i]*-log2((double)df[i]/dt);
    bits+=100*8; total_bits+=bits;
    uLongf zl=compressBound(n); static uint8_t zb[BS*2]; z_stream zs={0}; deflateInit2(&zs,6,Z_DEFLATED,-15,8,0); zs.next_in=buf; zs.avail_in=n; zs.next_out=zb; zs.avail_out=sizeof zb; deflate(&zs,Z_FINISH); zbytes+=zs.total_out; deflateEnd(&zs); (void)zl;
    inbytes+=n; nblk++;
  }
  printf("HB=%d W=%d T=%d MH=%d LAZY=%d MAXCMP=%d SHORTD=%d : est %.0f B  zlib6 %.0f B  ratio-vs-zlib %.4f  (comp ratio %.3f vs %.3f) tok/blk %.0f match%% %.1f\n",HB,W,T,MH,LAZY,MAXCMP,SHORTD,total_bits/8,zbytes,total_bits/8/zbytes,inbytes/(total_bits/8),inbytes/zbytes,(double)ntok/nblk,100.0*nmatch/ntok);
  return 0; }
