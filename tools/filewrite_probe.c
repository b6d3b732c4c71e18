// tools/filewrite_probe.c -- how fast do N bytes get into a file on tmpfs / disk: threads x {mmap stores, pwrite} x {as is, fallocate first, MAP_POPULATE}.
// usage: filewrite_probe <path> <MiB> <threads> <mode 0=mmap 1=pwrite> [pre 0|1=fallocate|2=populate] [block KiB]
#define _GNU_SOURCE
#include <fcntl.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <time.h>
#include <unistd.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+t.tv_nsec*1e-9;}
static size_t N; static int T, MODE, fd; static char *src, *map; static size_t BLK = 8u<<20;
static void *work(void *a){ long t=(long)a; size_t nb=N/BLK;
  for(size_t b=t;b<nb;b+=T){ size_t off=b*BLK;
    if(MODE==0) memcpy(map+off, src+(off% (256u<<20)), BLK);
    else { size_t done=0; while(done<BLK){ ssize_t w=pwrite(fd, src+(off%(256u<<20))+done, BLK-done, off+done); if(w<=0){perror("pw");exit(1);} done+=w; } }
  } return 0; }
int main(int c,char**v){ const char*path=v[1]; N=(size_t)atol(v[2])<<20; T=atoi(v[3]); MODE=atoi(v[4]); int pre=c>5?atoi(v[5]):0; if(c>6) BLK=(size_t)atol(v[6])<<10;
  src=malloc(256u<<20); memset(src,7,256u<<20);
  unlink(path); fd=open(path,O_RDWR|O_CREAT,0644);
  double t0=now();
  if(pre==1){ if(fallocate(fd,0,0,N)) perror("fallocate"); } else if(MODE==0) ftruncate(fd,N);
  double t1=now();
  if(MODE==0){ map=mmap(0,N,PROT_READ|PROT_WRITE,MAP_SHARED|(pre==2?MAP_POPULATE:0),fd,0); if(map==MAP_FAILED){perror("mmap");return 1;} }
  double t2=now();
  pthread_t th[256]; for(long t=0;t<T;++t) pthread_create(&th[t],0,work,(void*)t); for(int t=0;t<T;++t) pthread_join(th[t],0);
  double t3=now();
  printf("mode=%d T=%d pre=%d blk=%zuK: prealloc %.3f map %.3f write %.3f s  -> %.2f GB/s total %.3f\n",MODE,T,pre,BLK>>10,t1-t0,t2-t1,t3-t2,N/1e9/(t3-t0),t3-t0);
  close(fd); unlink(path); return 0; }
