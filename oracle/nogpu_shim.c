/* nogpu_shim -- measurement infrastructure of bench.py's `cpu_baseline` leg (never loaded by the product).
 *
 * The CPU baseline runs the oracle in single-threaded worker PROCESSES, one per host core.  They never need the GPU, but
 * on a ROCm build of torch the first backward makes the autograd engine ask every registered backend for its device count,
 * which initialises the HSA runtime and opens /dev/kfd -- whatever HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES say -- and a
 * GPU box admits only a handful of processes holding the device.  Preloaded into those workers (LD_PRELOAD, their
 * environment only), this shim makes the device nodes look absent: open() of /dev/kfd and /dev/dri/<node> fails with ENOENT,
 * the runtime reports zero devices, and "all cores" can mean all cores.
 *
 *   gcc -O2 -fPIC -shared -o libnogpu_shim.so nogpu_shim.c -ldl
 */
#define _GNU_SOURCE
#include <dlfcn.h>
#include <errno.h>
#include <fcntl.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <sys/types.h>

static int hidden(const char *p) { return p && (strcmp(p, "/dev/kfd") == 0 || strncmp(p, "/dev/dri/", 9) == 0); }

#define WRAP_OPEN(NAME)                                                        \
    int NAME(const char *path, int flags, ...) {                               \
        static int (*real)(const char *, int, ...);                            \
        mode_t mode = 0;                                                       \
        if (flags & (O_CREAT | O_TMPFILE)) {                                   \
            va_list ap;                                                        \
            va_start(ap, flags);                                               \
            mode = (mode_t)va_arg(ap, int);                                    \
            va_end(ap);                                                        \
        }                                                                      \
        if (hidden(path)) {                                                    \
            errno = ENOENT;                                                    \
            return -1;                                                         \
        }                                                                      \
        if (!real) real = (int (*)(const char *, int, ...))dlsym(RTLD_NEXT, #NAME); \
        return real(path, flags, mode);                                        \
    }
WRAP_OPEN(open)
WRAP_OPEN(open64)

#define WRAP_OPENAT(NAME)                                                      \
    int NAME(int dirfd, const char *path, int flags, ...) {                    \
        static int (*real)(int, const char *, int, ...);                       \
        mode_t mode = 0;                                                       \
        if (flags & (O_CREAT | O_TMPFILE)) {                                   \
            va_list ap;                                                        \
            va_start(ap, flags);                                               \
            mode = (mode_t)va_arg(ap, int);                                    \
            va_end(ap);                                                        \
        }                                                                      \
        if (hidden(path)) {                                                    \
            errno = ENOENT;                                                    \
            return -1;                                                         \
        }                                                                      \
        if (!real) real = (int (*)(int, const char *, int, ...))dlsym(RTLD_NEXT, #NAME); \
        return real(dirfd, path, flags, mode);                                 \
    }
WRAP_OPENAT(openat)
WRAP_OPENAT(openat64)

FILE *fopen(const char *path, const char *mode) {
    static FILE *(*real)(const char *, const char *);
    if (hidden(path)) {
        errno = ENOENT;
        return NULL;
    }
    if (!real) real = (FILE * (*)(const char *, const char *)) dlsym(RTLD_NEXT, "fopen");
    return real(path, mode);
}
