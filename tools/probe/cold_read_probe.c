#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <fcntl.h>
#include <unistd.h>
#include <pthread.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
static double now(){struct timespec t;clock_gettime(CLOCK_MONOTONIC,&t);return t.tv_sec+1e-9*t.tv_nsec;}
static int NF, T, MODE; static size_t SZ; static char* dst; static const char* dir;
static void* rd(void* a){ long id=(long)a; char p[256];
  for(int i=id;i<NF;i+=T){ snprintf(p,256,"%s/f%05d.bin",dir,i); int fd=open(p,O_RDONLY);
    char* d=dst+(size_t)i*SZ;
    if(MODE==0){ size_t o=0; while(o<SZ){ ssize_t r=pread(fd,d+o,SZ-o,o); if(r<=0)break; o+=r; } }
    else if(MODE==1){ void* m=mmap(0,SZ,PROT_READ,MAP_SHARED|MAP_POPULATE,fd,0); memcpy(d,m,SZ); munmap(m,SZ);}
    else if(MODE==2){ void* m=mmap(0,SZ,PROT_READ,MAP_SHARED,fd,0); memcpy(d,m,SZ); munmap(m,SZ);}
    else if(MODE==3){ void* m=mmap(0,SZ,PROT_READ,MAP_SHARED,fd,0); madvise(m,SZ,MADV_SEQUENTIAL); memcpy(d,m,SZ); munmap(m,SZ);}
    else if(MODE==4){ posix_fadvise(fd,0,0,POSIX_FADV_NOREUSE); void* m=mmap(0,SZ,PROT_READ,MAP_SHARED,fd,0); memcpy(d,m,SZ); munmap(m,SZ);}
    else if(MODE==5){ void* m=mmap(0,SZ,PROT_READ,MAP_PRIVATE,fd,0); madvise(m,SZ,MADV_SEQUENTIAL); memcpy(d,m,SZ); munmap(m,SZ);}
    close(fd);} return 0;}
static void wr(void){ char p[256]; char* b=malloc(SZ); for(size_t i=0;i<SZ;i++)b[i]=(char)(i*7);
  for(int i=0;i<NF;i++){ snprintf(p,256,"%s/f%05d.bin",dir,i); int fd=open(p,O_WRONLY|O_CREAT|O_TRUNC,0644); b[0]=(char)i; if(write(fd,b,SZ)!=(ssize_t)SZ)abort(); close(fd);} free(b);}
static double pass(void){ pthread_t th[64]; double t0=now(); for(long i=0;i<T;i++)pthread_create(&th[i],0,rd,(void*)i); for(int i=0;i<T;i++)pthread_join(th[i],0); return now()-t0;}
int main(int c,char**v){ dir=v[1]; NF=atoi(v[2]); T=atoi(v[3]); SZ=2880044; dst=malloc((size_t)NF*SZ); memset(dst,1,(size_t)NF*SZ);
  for(MODE=0;MODE<6;MODE++){ wr(); double a=pass(), b=pass(), cc=pass(); double gb=NF*(double)SZ/1e9;
    printf("mode %d (%s): first %.3f s (%.1f GB/s) second %.3f (%.1f) third %.3f (%.1f)\n",MODE,MODE==0?"pread":MODE==1?"mmap populate":MODE==2?"mmap":MODE==3?"mmap+MADV_SEQUENTIAL":MODE==4?"NOREUSE+mmap":"mmap private+SEQ",a,gb/a,b,gb/b,cc,gb/cc);} 
  char cmd[300]; snprintf(cmd,300,"rm -rf %s",dir); return system(cmd);}
