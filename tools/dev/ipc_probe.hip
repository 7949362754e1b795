// ipc_probe.hip -- does HIP IPC (memory handles + interprocess events) order a kernel in one process behind a kernel of
// another process on this pool, with both processes on ONE GPU?  Parent forks before either touches the GPU.
//   hipcc --offload-arch=gfx950 -O2 -o ipc_probe ipc_probe.hip && ./ipc_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sys/mman.h>
#include <sys/wait.h>
#include <unistd.h>

#define OK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "[%d] %s -> %s (line %d)\n", getpid(), #e, hipGetErrorString(r_), __LINE__); _exit(3); } } while (0)

__global__ void k_slow_write(volatile long long* p, long long v, int spin)
{
    long long t0 = wall_clock64();
    while (wall_clock64() - t0 < spin) {}
    p[0] = v;
}
__global__ void k_read(const volatile long long* p, long long* out) { out[0] = p[0]; }

struct Shared {
    hipIpcMemHandle_t mem;
    hipIpcEventHandle_t ev[2];
    volatile int ready;
    volatile long long seq;         // deposits the writer has recorded
    volatile long long done;        // ... the reader has checked
    volatile int fail;
};

int main()
{
    Shared* sh = (Shared*)mmap(nullptr, sizeof(Shared), PROT_READ | PROT_WRITE, MAP_SHARED | MAP_ANONYMOUS, -1, 0);
    memset(sh, 0, sizeof(Shared));
    const int N = 200;
    pid_t pid = fork();
    if (pid != 0) {                 // writer
        long long* buf = nullptr;
        hipEvent_t ev[2];
        hipStream_t st;
        OK(hipSetDevice(0));
        OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
        OK(hipMalloc((void**)&buf, 4096));
        OK(hipMemset(buf, 0, 4096));
        OK(hipIpcGetMemHandle(&sh->mem, buf));
        for (int i = 0; i < 2; ++i) {
            OK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming | hipEventInterprocess));
            OK(hipEventRecord(ev[i], st));
            OK(hipStreamSynchronize(st));
            { hipError_t r = hipIpcGetEventHandle(&sh->ev[i], ev[i]); if (r != hipSuccess) { fprintf(stderr, "hipIpcGetEventHandle: %s; /dev/shm writable: %d\n", hipGetErrorString(r), access("/dev/shm", W_OK) == 0); sh->fail = 1; sh->ready = 1; _exit(3); } }
        }
        __sync_synchronize();
        sh->ready = 1;
        for (long long k = 1; k <= N; ++k) {
            while (sh->done < k - 1 && !sh->fail) {}      // reader has consumed k - 1 (one buffer: no overwrite before it is read)
            hipLaunchKernelGGL(k_slow_write, dim3(1), dim3(1), 0, st, buf, k, 5000 /* ~50 us at 100 MHz */);
            OK(hipEventRecord(ev[k & 1], st));
            __sync_synchronize();
            sh->seq = k;
        }
        OK(hipStreamSynchronize(st));
        int status = 0;
        waitpid(pid, &status, 0);
        printf("writer done; reader exit %d, fail %d\n", WEXITSTATUS(status), sh->fail);
        return sh->fail || WEXITSTATUS(status);
    }
    // reader
    while (!sh->ready) {}
    if (sh->fail) _exit(4);
    OK(hipSetDevice(0));
    hipStream_t st;
    OK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
    long long* buf = nullptr;
    OK(hipIpcOpenMemHandle((void**)&buf, sh->mem, hipIpcMemLazyEnablePeerAccess));
    hipEvent_t ev[2];
    for (int i = 0; i < 2; ++i) OK(hipIpcOpenEventHandle(&ev[i], sh->ev[i]));
    long long* out = nullptr;
    OK(hipHostMalloc((void**)&out, 64, hipHostMallocDefault));
    int bad = 0;
    double waitUs = 0;
    for (long long k = 1; k <= N; ++k) {
        while (sh->seq < k) {}
        auto t0 = std::chrono::steady_clock::now();
        OK(hipStreamWaitEvent(st, ev[k & 1], 0));
        hipLaunchKernelGGL(k_read, dim3(1), dim3(1), 0, st, buf, out);
        OK(hipStreamSynchronize(st));
        waitUs += std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (out[0] != k) { if (bad < 5) fprintf(stderr, "reader: deposit %lld saw %lld\n", k, out[0]); ++bad; }
        __sync_synchronize();
        sh->done = k;
    }
    printf("reader: %d of %d reads saw a stale value; wait+read+sync %.1f us on average (the writer's kernel spins ~50 us)\n", bad, N, waitUs / N);
    if (bad) sh->fail = 1;
    _exit(bad ? 1 : 0);
}
