/* walk_length_predictor.c — research tool (not product, not oracle): how well does the distance to the hash-chain head predict the
 * length of a FindLongestMatch walk (C/DeflaterEngine.cs:474-612)?  Input for the two-pass hand-out of k_match4o.
 *   gcc -O2 -o /tmp/pred tools/walk_length_predictor.c && /tmp/pred sample.bin <bytes> <max_chain>
 * (the hash here is a stand-in with the reference's shape — three bytes into 15 bits — not its exact constants) */
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include <string.h>
// how well does the first hop distance predict the length of a FindLongestMatch walk?
int main(int argc,char**argv){ FILE*f=fopen(argv[1],"rb"); size_t n=atol(argv[2]); uint8_t*d=malloc(n+8); n=fread(d,1,n,f);
 uint16_t*link=calloc(n,2); int32_t*head=malloc(32768*4); for(int i=0;i<32768;i++)head[i]=-1;
 for(size_t q=0;q+2<n;q++){ uint32_t h=((d[q]<<10)^(d[q+1]<<5)^d[q+2])&32767; if(head[h]>=0 && q-head[h]<=32767) link[q]=q-head[h]; head[h]=q; }
 int MC=atoi(argv[3]);
 uint8_t*L=calloc(n,1);
 for(size_t q=65536;q<n;q++){ size_t c=q; int k=0; while(k<MC){ uint32_t l=link[c]; if(!l)break; c-=l; if(q-c>32505)break; k++; } L[q]=k; }
 int ths[]={256,1024,4096,8192,16384};
 for(int t=0;t<5;t++){ double a=0,b=0; size_t na=0,nb=0,longA=0,longB=0; for(size_t q=65536;q<n;q++){ uint32_t l0=link[q]; int A=l0&&l0<ths[t]; if(A){a+=L[q];na++;longA+=L[q]>=MC*3/4;}else{b+=L[q];nb++;longB+=L[q]>=MC*3/4;} }
  printf("TH %5d: class A %.1f%% avg walk %.1f long(>=%d) %.1f%% | class B %.1f%% avg walk %.1f long %.2f%%\n",ths[t],100.0*na/(na+nb),a/na,MC*3/4,100.0*longA/na,100.0*nb/(na+nb),b/nb,100.0*longB/nb); }
 return 0; }
