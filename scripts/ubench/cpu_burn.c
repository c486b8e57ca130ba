// cpu_burn.c -- busy threads pinned to a CPU list, for the host-contention test of bench.py (tests/test_gpu_bench.py):
// stands for the seven OTHER ranks of an 8-GPU run on this node's host cores.
//   cpu_burn <seconds> <cpu> [<cpu> ...]      one spinning thread per listed CPU, exits after <seconds>
// build: gcc -O2 -pthread scripts/ubench/cpu_burn.c -o cpu_burn
#define _GNU_SOURCE
#include <pthread.h>
#include <sched.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>

static volatile int stop_flag = 0;

static void* spin(void* arg) {
    const int cpu = (int)(long)arg;
    cpu_set_t set;
    CPU_ZERO(&set);
    CPU_SET(cpu, &set);
    pthread_setaffinity_np(pthread_self(), sizeof(set), &set);
    volatile unsigned long x = 1;
    while (!stop_flag) x = x * 6364136223846793005ul + 1442695040888963407ul;
    return (void*)x;
}

int main(int argc, char** argv) {
    if (argc < 3) {
        fprintf(stderr, "usage: %s <seconds> <cpu>...\n", argv[0]);
        return 2;
    }
    const double seconds = atof(argv[1]);
    const int n = argc - 2;
    pthread_t* th = (pthread_t*)calloc((size_t)n, sizeof(pthread_t));
    for (int i = 0; i < n; ++i) pthread_create(&th[i], NULL, spin, (void*)(long)atoi(argv[2 + i]));
    printf("burning %d cpus\n", n);
    fflush(stdout);
    struct timespec ts = {(time_t)seconds, (long)((seconds - (long)seconds) * 1e9)};
    nanosleep(&ts, NULL);
    stop_flag = 1;
    for (int i = 0; i < n; ++i) pthread_join(th[i], NULL);
    free(th);
    return 0;
}
