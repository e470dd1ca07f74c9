/* lzsim2.c -- design-space explorer for the GPU deflate match finder, second generation (NOT product code).
 * Model A = the kernel of rounds 1-3: every position compares itself with the W most recent earlier positions of its hash bucket (+ distance 1) --
 *   all candidates over LOCKSTEP bytes, then only the nearest survivor to the end.
 * Model B = "heads + prefix maximum": a (position, candidate) pair is looked at only when the bytes BEFORE the two differ (or at a chunk's first
 *   position): it is the head of a maximal repeat; its run is measured once (cap CAP bytes) and every later position on the same diagonal inherits
 *   "run end - position" through a prefix maximum over run ends.  Same table, same insertion order, same parse.
 * Reports the entropy-estimated size against zlib level 6 and the work each model does.
 * usage: lzsim2 file WAYS LAZY(0|1|2) MODEL(A|B) [CAP] [LOCKSTEP0 LOCKSTEP1]
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <zlib.h>
#define BS 65280
#define T 256
#define HB 9
static int lsym(int len){ static const int base[29]={3,4,5,6,7,8,9,10,11,13,15,17,19,23,27,31,35,43,51,59,67,83,99,115,131,163,195,227,258}; int s=28; while(base[s]>len) s--; return s; }
static int lext(int s){ static const int e[29]={0,0,0,0,0,0,0,0,1,1,1,1,2,2,2,2,3,3,3,3,4,4,4,4,5,5,5,5,0}; return e[s]; }
static int dsym(int d){ static const int base[30]={1,2,3,4,5,7,9,13,17,25,33,49,65,97,129,193,257,385,513,769,1025,1537,2049,3073,4097,6145,8193,12289,16385,24577}; int s=29; while(base[s]>d) s--; return s; }
static int dext(int s){ return s<4?0:(s-2)>>1; }
int main(int argc,char**argv){
  FILE*f=fopen(argv[1],"rb"); int W=atoi(argv[2]),LAZY=atoi(argv[3]); char model=argv[4][0]; int CAP=argc>5?atoi(argv[5]):514;
  int LS0=argc>6?atoi(argv[6]):32, LS1=argc>7?atoi(argv[7]):24;
  static uint8_t buf[BS+64]; size_t n; double total_bits=0, zbytes=0, inbytes=0; long nblk=0, ntok=0, nmatch=0;
  int HS=1<<HB; uint16_t*tab=malloc(sizeof(uint16_t)*HS*12); uint8_t*cnt=malloc(HS);
  int *mlen=calloc(BS+T+4,sizeof(int)), *mdist=calloc(BS+T+4,sizeof(int));
  long pairs=0, heads=0, rounds16=0, positions=0, waves=0, maxheads_sum=0, head_ge3=0; long hist[16]={0};
  while((n=fread(buf,1,BS,f))>0){
    memset(buf+n,0,64); memset(tab,0xff,sizeof(uint16_t)*HS*12); memset(cnt,0,HS);
    long cur_end=0; int cur_d=0;                                   /* model B: the carried best diagonal */
    for(size_t c0=0;c0<n;c0+=T){
      size_t c1=c0+T>n?n:c0+T;
      int wave_max=0;
      for(size_t p=c0;p<c1;p++){
        int best=0,bd=0; size_t maxl=n-p>258?258:n-p; int hashable=p+4<=n; int nheads=0;
        long kend=0; int kd=0;
        if(hashable){
          uint32_t v; memcpy(&v,buf+p,4); uint32_t h=(v*2654435761u)>>(32-HB);
          int cand[13], nc=0;
          for(int w=0;w<W;w++){ uint16_t c=tab[h*12+w]; if(c==0xffff||p-c>32768) continue; cand[nc++]=c; }
          if(p>=1) cand[nc++]=p-1;
          positions++; pairs+=nc;
          if(model=='A'){
            /* all candidates over the lock-step bytes (group 0: first 8 ways + d1, group 1: the rest), nearest survivor of each group to the end */
            for(int g=0;g<2;g++){
              int lo=g==0?0:8, hi=g==0?(W<8?W:8):W; int ls=g==0?LS0:LS1; if(W<=8) ls=LS0;
              int idx[16],k=0; for(int i=0;i<nc;i++){ int istab = i<nc-(p>=1); int way=i; if(istab){ if(way>=lo&&way<hi) idx[k++]=i; } else if(g==0) idx[k++]=i; }
              /* (ways are counted over valid entries here; the kernel counts table slots -- same thing once a bucket is full) */
              int len[16]; int near=-1; int nd=1<<30;
              for(int j=0;j<k;j++){ int c=cand[idx[j]]; size_t l=0; size_t ml=maxl<(size_t)ls?maxl:ls; while(l<ml&&buf[c+l]==buf[p+l]) l++; len[j]=l; rounds16+=(l+15)/16;
                if(l>=(size_t)ls&&l<maxl&&(int)(p-c)<nd){ nd=p-c; near=j; } }
              if(near>=0){ int c=cand[idx[near]]; size_t l=len[near]; while(l<maxl&&buf[c+l]==buf[p+l]) l++; rounds16+=(l-len[near]+15)/16; len[near]=l; }
              for(int j=0;j<k;j++){ int d=p-cand[idx[j]]; if(len[j]>=3&&(len[j]>best||(len[j]==best&&d<bd))){best=len[j];bd=d;} }
              if(W<=8) break;
            }
          } else {
            for(int i=0;i<nc;i++){
              int c=cand[i];
              int head = p==c0 || c==0 || buf[p-1]!=buf[c-1];
              if(!head) continue;
              nheads++; heads++;
              size_t ml=n-p<(size_t)CAP?n-p:(size_t)CAP; size_t l=0; while(l<ml&&buf[c+l]==buf[p+l]) l++;
              rounds16+=l/16+1;
              if(l>=3){ head_ge3++; long e=p+l; int d=p-c; if(e>kend||(e==kend&&d<kd)){kend=e;kd=d;} }
            }
            hist[nheads>15?15:nheads]++;
            if(nheads>wave_max) wave_max=nheads;
          }
        }
        if(model=='B'){
          if(kend>cur_end||(kend==cur_end&&kend>0&&kd<cur_d)){cur_end=kend;cur_d=kd;}
          if(hashable&&cur_end>(long)p){ long l=cur_end-p; if(l>(long)maxl) l=maxl; best=l; bd=cur_d; }
          if(((p-c0)&63)==63||p+1==c1){ waves++; maxheads_sum+=wave_max; wave_max=0; }
        }
        if(best<3||(best==3&&bd>4096)||(LAZY>=2&&best==4&&bd>2048)) best=0;
        mlen[p]=best; mdist[p]=bd;
      }
      for(size_t p=c0;p<c1&&p+4<=n;p++){ uint32_t v; memcpy(&v,buf+p,4); uint32_t h=(v*2654435761u)>>(32-HB); tab[h*12+(cnt[h]++%W)]=p; }
    }
    /* parse: per chunk, the last positions cannot look ahead (mlen beyond the chunk reads as 0) */
    long lf[286]={0},df[30]={0}; double extra=0; size_t p=0;
    while(p<n){
      int l=mlen[p]; size_t cend=(p/T+1)*T;
      int n1=p+1<cend&&p+1<n?mlen[p+1]:0, n2=p+2<cend&&p+2<n?mlen[p+2]:0;
      if(l&&LAZY>=1&&n1>l) l=0;
      if(l&&LAZY>=2&&n2>l+1) l=0;
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
  printf("model %c W=%d LAZY=%d CAP=%d : est %.0f B  zlib6 %.0f B  size-vs-zlib6 %.4f  tok/blk %.0f match%% %.1f | pairs/pos %.2f rounds16/pos %.2f",model,W,LAZY,CAP,total_bits/8,zbytes,total_bits/8/zbytes,(double)ntok/nblk,100.0*nmatch/ntok,(double)pairs/positions,(double)rounds16/positions);
  if(model=='B'){ printf(" heads/pos %.3f (>=3 bytes: %.3f) mean of per-wave max heads %.2f\n   heads per position histogram:",(double)heads/positions,(double)head_ge3/positions,(double)maxheads_sum/waves); for(int i=0;i<16;i++) printf(" %d:%.3f",i,(double)hist[i]/positions); }
  printf("\n");
  return 0; }
