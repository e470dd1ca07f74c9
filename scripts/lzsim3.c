/* lzsim3.c -- design-space explorer for the GPU deflate match finder, third generation (NOT product code).
 * Models a TIERED search: every position compares itself with the W0 most recent entries of its hash bucket (+ distance 1) over LS0 bytes, the
 * nearest survivor is extended; optionally a position inherits "length - 1 at the same distance" from its left neighbour (INH); only positions
 * whose best match is then shorter than T look at the remaining W - W0 ways (over LS1 bytes, nearest survivor extended).  Reports the
 * entropy-estimated size against zlib level 6, the fraction of positions that need the second tier, and the number of 64-lane groups the
 * second tier fills per 256-position chunk once its positions are compacted.
 * usage: lzsim3 file W W0 T INH(0|1) LAZY [LS0 LS1 WINDOW RECENCY(0|1) NBLK]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <zlib.h>
#define BS 65280
#define CH 256
#define HB 9
static int lsym(int len){ static const int base[29]={3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258}; int s=28; while(base[s]>len) s--; return s; }
static int lext(int s){ static const int e[29]={0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0}; return e[s]; }
static int dsym(int d){ static const int base[30]={1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577}; int s=29; while(base[s]>d) s--; return s; }
static int dext(int s){ return s<4?0:(s-2)>>1; }
static uint8_t buf[BS+600];
static size_t n;
/* one tier: candidates cand[0..k) over ls bytes, nearest full survivor extended; updates best/bd */
static int NOEXT=0;
static void tier(size_t p,const int*cand,int k,int ls,size_t maxl,int*best,int*bd,long*cmp,long*ext16){
  int len[16]; int near=-1,nd=1<<30;
  for(int j=0;j<k;j++){ int c=cand[j]; size_t l=0; size_t ml=maxl<(size_t)ls?maxl:(size_t)ls; while(l<ml&&buf[c+l]==buf[p+l]) l++; len[j]=l; (*cmp)++;
    if(l>=(size_t)ls&&l<maxl&&(int)(p-c)<nd){ nd=p-c; near=j; } }
  if(near>=0&&!NOEXT){ int c=cand[near]; size_t l=len[near]; while(l<maxl&&buf[c+l]==buf[p+l]) l++; *ext16+=(l-len[near]+15)/16; len[near]=l; }
  for(int j=0;j<k;j++){ int d=p-cand[j]; if(len[j]>=3&&(len[j]>*best||(len[j]==*best&&d<*bd))){*best=len[j];*bd=d;} }
}
int main(int argc,char**argv){ int INS=getenv("INS")?atoi(getenv("INS")):0, INSL=getenv("INSL")?atoi(getenv("INSL")):16, INSK=getenv("INSK")?atoi(getenv("INSK")):1;
  if(argc<7){ fprintf(stderr,"usage\n"); return 1; }
  FILE*f=fopen(argv[1],"rb"); int W=atoi(argv[2]),W0=atoi(argv[3]),T=atoi(argv[4]),INH=atoi(argv[5]),LAZY=atoi(argv[6]);
  int LS0=argc>7?atoi(argv[7]):32, LS1=argc>8?atoi(argv[8]):24, WIN=argc>9?atoi(argv[9]):32768, REC=argc>10?atoi(argv[10]):1; long NBLK=argc>11?atol(argv[11]):1000000; NOEXT=(INH==3); int LOOK=argc>12?atoi(argv[12]):2; int TU=argc>13?atoi(argv[13]):0; int ITER=argc>14?atoi(argv[14]):1;
  double total_bits=0, zbytes=0, inbytes=0; long nblk=0, ntok=0, nmatch=0;
  int HS=1<<HB; uint16_t*tab=malloc(sizeof(uint16_t)*HS*16); uint32_t*cnt=malloc(HS*4);
  int *mlen=calloc(BS+CH+4,sizeof(int)), *mdist=calloc(BS+CH+4,sizeof(int));
  long positions=0, deep=0, deepgroups=0, chunks=0, cmp0=0, cmp1=0, ext0=0, ext1=0, tokstart_deep=0, inh_used=0;
  while(nblk<NBLK&&(n=fread(buf,1,BS,f))>0){
    memset(buf+n,0,64); memset(tab,0xff,sizeof(uint16_t)*HS*16); memset(cnt,0,HS*4);
    static uint8_t isdeep[BS+CH]; size_t carry_in=0; size_t pcarry=0;
    for(size_t c0=0;c0<n;c0+=CH){
      size_t c1=c0+CH>n?n:c0+CH; int ndeep=0;
      int pb=0,pd=0; static int sbest[CH],sbd[CH];                                    /* left neighbour's best (for inheritance), reset per chunk */
      for(size_t p=c0;p<c1;p++){
        int best=0,bd=0; size_t maxl=n-p>258?258:n-p; int hashable=p+4<=n; isdeep[p]=0; sbest[p-c0]=0; sbd[p-c0]=0;
        if(hashable){
          uint32_t v; memcpy(&v,buf+p,4); uint32_t h=(v*2654435761u)>>(32-HB);
          int c0list[16],k0=0,c1list[16],k1=0;
          for(int w=0;w<W;w++){
            int slot = REC ? (int)((cnt[h]+W*1024-1-w)%W) : w;      /* recency order: w-th most recent */
            uint16_t c=tab[h*16+slot]; if(c==0xffff||(int)(p-c)>WIN) continue;
            if(w<W0) c0list[k0++]=c; else c1list[k1++]=c;
          }
          if(p>=1) c0list[k0++]=p-1;
          positions++;
          tier(p,c0list,k0,LS0,maxl,&best,&bd,&cmp0,&ext0);
          if(INH==1&&pb>=4&&p>c0){ int il=pb-1; if(il>(int)maxl) il=maxl; if(il>best||(il==best&&pd<bd)){ best=il; bd=pd; inh_used++; } }
          sbest[p-c0]=best; sbd[p-c0]=bd; if(INH<2&&best<T&&k1>0){ deep++; ndeep++; isdeep[p]=1; tier(p,c1list,k1,LS1,maxl,&best,&bd,&cmp1,&ext1); }
        }
        pb=best; pd=bd;
        if(best<3||(best==3&&bd>4096)||(LAZY>=2&&best==4&&bd>2048)) best=0;
        mlen[p]=best; mdist[p]=bd;
      }
      if(INH==2){
        /* scheme S: preliminary parse of this chunk with tier-0 lengths; entry = carry from the final parse of the previous chunk */
        static uint8_t want[CH+4]; memset(want,0,sizeof want); size_t q;
        for(size_t p=c0;p<c1;p++) if(sbest[p-c0]<TU) want[p-c0]=1;
        for(int it=0;it<ITER;it++){
        q=c0+carry_in; 
        while(q<c1){ int l=mlen[q]; int n1=q+1<c1?mlen[q+1]:0,n2=q+2<c1?mlen[q+2]:0; if(l&&LAZY>=1&&n1>l) l=0; if(l&&LAZY>=2&&n2>l+1) l=0;
          want[q-c0]=1; if(LOOK>=1) want[q-c0+1]=1; if(LOOK>=2) want[q-c0+2]=1; q+= l?l:1; }
        for(size_t p=c0;p<c1;p++) if(want[p-c0]&&!isdeep[p]&&p+4<=n&&sbest[p-c0]<T){
          uint32_t v; memcpy(&v,buf+p,4); uint32_t h=(v*2654435761u)>>(32-HB); int c1list[16],k1=0; size_t maxl=n-p>258?258:n-p;
          for(int w=W0;w<W;w++){ int slot = REC ? (int)((cnt[h]+W*1024-1-w)%W) : w; uint16_t c=tab[h*16+slot]; if(c==0xffff||(int)(p-c)>WIN) continue; c1list[k1++]=c; }
          if(k1>0){ int best=sbest[p-c0],bd=sbd[p-c0]; deep++; ndeep++; isdeep[p]=1; tier(p,c1list,k1,LS1,maxl,&best,&bd,&cmp1,&ext1);
            if(best<3||(best==3&&bd>4096)||(LAZY>=2&&best==4&&bd>2048)) best=0; mlen[p]=best; mdist[p]=bd; } }
        }
        /* final parse of the chunk to get the carry */
        q=c0+carry_in; while(q<c1){ int l=mlen[q]; int n1=q+1<c1?mlen[q+1]:0,n2=q+2<c1?mlen[q+2]:0; if(l&&LAZY>=1&&n1>l) l=0; if(l&&LAZY>=2&&n2>l+1) l=0; q+= l?l:1; }
        carry_in=q-c1;
      }

      if(INH==3){
        static uint8_t want[CH+4]; memset(want,0,sizeof want);
        /* tier-1 results so far: sbest capped (no extension): recompute capped values */
        for(int seg=0;seg<4;seg++){ size_t s0=c0+64*seg, s1=s0+64>c1?c1:s0+64; if(s0>=c1) break;
          size_t q= seg==0? c0+carry_in : s0;
          while(q<s1){ int l=mlen[q]; int n1=q+1<c1?mlen[q+1]:0,n2=q+2<c1?mlen[q+2]:0; if(l&&LAZY>=1&&n1>l) l=0; if(l&&LAZY>=2&&n2>l+1) l=0;
            want[q-c0]=1; if(LOOK>=1&&q+1<s1) want[q-c0+1]=1; q+= l?l:1; } }
        for(size_t p=c0;p<c1;p++) if(p+4<=n){
          uint32_t v; memcpy(&v,buf+p,4); uint32_t h=(v*2654435761u)>>(32-HB); size_t maxl=n-p>258?258:n-p;
          int best=sbest[p-c0],bd=sbd[p-c0];   /* capped tier-1 result: best<=LS0; if best==LS0 (and maxl>LS0), bd = nearest survivor */
          int xb=0,xd=0; if(TU==1&&best==LS0&&(int)maxl>LS0){ size_t c=p-bd; size_t l=best; while(l<maxl&&buf[c+l]==buf[p+l]) l++; ext0+=(l-best+15)/16; xb=l; xd=bd; best=0; bd=0; }
          if(want[p-c0]){
            int k1=0; deep++; ndeep++; isdeep[p]=1;
            for(int w=W0;w<W;w++){ int slot = REC ? (int)((cnt[h]+W*1024-1-w)%W) : w; uint16_t c=tab[h*16+slot]; if(c==0xffff||(int)(p-c)>WIN) continue; k1++;
              size_t l=0; size_t ml=maxl<(size_t)LS0?maxl:(size_t)LS0; while(l<ml&&buf[c+l]==buf[p+l]) l++; cmp1++; int d=p-c;
              if((int)l>=3&&((int)l>best||((int)l==best&&d<bd))){best=l;bd=d;} }
          }
          if(best==LS0&&(int)maxl>LS0){ size_t c=p-bd; size_t l=best; while(l<maxl&&buf[c+l]==buf[p+l]) l++; ext1+=(l-best+15)/16; best=l; }
          if(xb>best||(xb==best&&xd<bd)){best=xb;bd=xd;}
          if(best<3||(best==3&&bd>4096)||(LAZY>=2&&best==4&&bd>2048)) best=0; mlen[p]=best; mdist[p]=bd;
        }
        size_t q=c0+carry_in; while(q<c1){ int l=mlen[q]; int n1=q+1<c1?mlen[q+1]:0,n2=q+2<c1?mlen[q+2]:0; if(l&&LAZY>=1&&n1>l) l=0; if(l&&LAZY>=2&&n2>l+1) l=0; q+= l?l:1; }
        carry_in=q-c1;
      }
      chunks++; deepgroups+=(ndeep+63)/64;
      { /* parse this chunk (mode 0 path: INH<2) to know the token starts; pcarry = chunk-relative entry */
        static uint8_t skip[CH]; memset(skip,0,sizeof skip);
        if(INS){ size_t q=c0+pcarry; if(INS==1) memset(skip,1,sizeof skip);
          while(q<c1){ int l=mlen[q]; int n1=q+1<c1?mlen[q+1]:0,n2=q+2<c1?mlen[q+2]:0; if(l&&LAZY>=1&&n1>l) l=0; if(l&&LAZY>=2&&n2>l+1) l=0;
            if(INS==1) skip[q-c0]=0;
            if(INS==2&&l>=INSL){ for(int k=INSK;k<l&&q+k<c1;k++) skip[q+k-c0]=1; }
            q+= l?l:1; }
          pcarry=q-c1; }
        for(size_t p=c0;p<c1&&p+4<=n;p++){ if(skip[p-c0]) continue; uint32_t v; memcpy(&v,buf+p,4); uint32_t h=(v*2654435761u)>>(32-HB); tab[h*16+(cnt[h]++%W)]=p; } }
    }
    long lf[286]={0},df[30]={0}; double extra=0; size_t p=0;
    while(p<n){
      int l=mlen[p]; size_t cend=(p/CH+1)*CH;
      int n1=p+1<cend&&p+1<n?mlen[p+1]:0, n2=p+2<cend&&p+2<n?mlen[p+2]:0;
      if(l&&LAZY>=1&&n1>l) l=0;
      if(l&&LAZY>=2&&n2>l+1) l=0;
      tokstart_deep+=isdeep[p];
      if(l){ int s=lsym(l); lf[257+s]++; extra+=lext(s); int ds=dsym(mdist[p]); df[ds]++; extra+=dext(ds); p+=l; nmatch++; }
      else { lf[buf[p]]++; p++; }
      ntok++;
    }
    lf[256]=1; double bits=extra; long lt=0,dt=0; for(int i=0;i<286;i++) lt+=lf[i]; for(int i=0;i<30;i++) dt+=df[i];
    for(int i=0;i<286;i++) if(lf[i]) bits+=lf[i]*-log2((double)lf[i]/lt);
    for(int i=0;i<30;i++) if(df[i]) bits+=df[i]*-log2((double)df[i]/dt);
    bits+=100*8; total_bits+=bits;
    static uint8_t zb[BS*2]; z_stream zs={0}; deflateInit2(&zs,6,Z_DEFLATED,-15,8,0); zs.next_in=buf; zs.avail_in=n; zs.next_out=zb; zs.avail_out=sizeof zb; deflate(&zs,Z_FINISH); zbytes+=zs.total_out; deflateEnd(&zs);
    inbytes+=n; nblk++;
  }
  printf("W=%d W0=%d T=%d INH=%d LAZY=%d LS=%d/%d WIN=%d REC=%d : size-vs-zlib6 %.4f ratio %.3f tok/pos %.3f | deep/pos %.3f deepgroups/chunk %.2f (of 4) cmp0/pos %.2f cmp1/pos %.2f ext16 %.2f+%.2f inh %.3f tokstart-deep %.3f\n",
    W,W0,T,INH,LAZY,LS0,LS1,WIN,REC,total_bits/8/zbytes,inbytes/(total_bits/8),(double)ntok/positions,(double)deep/positions,(double)deepgroups/chunks,(double)cmp0/positions,(double)cmp1/positions,(double)ext0/positions,(double)ext1/positions,(double)inh_used/positions,(double)tokstart_deep/ntok);
  return 0; }
