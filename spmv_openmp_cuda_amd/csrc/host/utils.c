/*
 * utils.c -- vector I/O, environment configuration, the parity gate and timing
 * statistics (host side).  Behaviour follows the reference's
 * src/commons/utils.c:135-393; written from scratch.  Differences: the diff
 * gate and the random vector treat NaN as an error instead of letting it
 * through (see utils.h), file readers release everything on their error paths.
 */
#include <errno.h>
#include <fcntl.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/stat.h>
#include <unistd.h>

#include <dlfcn.h>
#include <zlib.h>

#include "utils.h"

int urndFd = -1;

int init_urndfd(void) {
    if ((urndFd = open(DRNG_DEVFILE, O_RDONLY)) < 0) { perror("open DRNG_DEVFILE"); return EXIT_FAILURE; }
    return EXIT_SUCCESS;
}

static int readFully(int fd, void* dst, size_t count) {
    size_t done = 0;
    while (done < count) {
        ssize_t rd = read(fd, (char*)dst + done, count - done);
        if (rd < 0) { if (errno == EINTR) continue; perror("read"); return EXIT_FAILURE; }
        if (rd == 0) return EXIT_FAILURE;
        done += (size_t)rd;
    }
    return EXIT_SUCCESS;
}
static int writeFully(int fd, const void* src, size_t count) {
    size_t done = 0;
    while (done < count) {
        ssize_t wr = write(fd, (const char*)src + done, count - done);
        if (wr < 0) { if (errno == EINTR) continue; perror("write"); return EXIT_FAILURE; }
        done += (size_t)wr;
    }
    return EXIT_SUCCESS;
}

int writeDoubleVector(char* fpath, double* v, ulong size) {
    int fd = open(fpath, O_WRONLY | O_CREAT | O_TRUNC, S_IRWXU);
    if (fd < 0) { perror("open outFd failed "); return EXIT_FAILURE; }
    int rc = writeFully(fd, v, size * sizeof *v);
    if (close(fd)) { perror("close errd\n"); rc = EXIT_FAILURE; }
    return rc;
}

int writeDoubleVectorAsStr(char* fpath, double* v, ulong size) {
    FILE* fp = fopen(fpath, "w");
    if (!fp) { perror("fopen vector file write"); return EXIT_FAILURE; }
    int rc = EXIT_SUCCESS;
    for (ulong i = 0; i < size && !rc; ++i)
        if (fprintf(fp, DOUBLE_STR_FORMAT, v[i]) < 0) { ERRPRINT("fprintf to out vector file errd\n"); rc = EXIT_FAILURE; }
    if (fclose(fp) == EOF) { perror("fclose errd\n"); rc = EXIT_FAILURE; }
    return rc;
}

double* readDoubleVector(char* fpath, ulong* size) {
    FILE* fp = fopen(fpath, "rb");
    if (!fp) { perror("fopen vector file"); return NULL; }
    ulong cap = *size ? *size : RNDVECTORSIZE, n = 0;
    double* out = malloc(cap * sizeof *out);
    if (!out) { ERRPRINT("vector read malloc fail for file\n"); fclose(fp); return NULL; }
    for (;;) {
        if (n == cap) {
            cap *= VECTOR_STEP_REALLOC;
            double* t = realloc(out, cap * sizeof *out);
            if (!t) { ERRPRINTS("realloc errd to ~~ %lu MB\n", (cap * sizeof *out) >> 20); free(out); fclose(fp); return NULL; }
            out = t;
        }
        size_t got = fread(out + n, sizeof *out, cap - n, fp);
        n += got;
        if (got == 0) break;
    }
    if (ferror(fp)) { ERRPRINT("fread errd\n"); free(out); fclose(fp); return NULL; }
    fclose(fp);
    if (n == 0) { ERRPRINT("empty vector file\n"); free(out); return NULL; }
    *size = n;
    double* t = realloc(out, n * sizeof *out);
    return t ? t : out;
}

double* readDoubleVectorStr(char* fpath, ulong* size) {
    FILE* fp = fopen(fpath, "r");
    if (!fp) { perror("fopen vector file"); return NULL; }
    ulong cap = *size ? *size : RNDVECTORSIZE, n = 0;
    double* out = malloc(cap * sizeof *out);
    if (!out) { ERRPRINT("vector read malloc fail for file\n"); fclose(fp); return NULL; }
    double v;
    while (fscanf(fp, "%le", &v) == 1) {
        if (n == cap) {
            cap *= VECTOR_STEP_REALLOC;
            double* t = realloc(out, cap * sizeof *out);
            if (!t) { ERRPRINT("realloc errd\n"); free(out); fclose(fp); return NULL; }
            out = t;
        }
        out[n++] = v;
    }
    if (ferror(fp)) { perror("invalid fscanf"); free(out); fclose(fp); return NULL; }
    fclose(fp);
    if (n == 0) { ERRPRINT("empty vector file\n"); free(out); return NULL; }
    *size = n;
    return out;
}

int getConfig(CONFIG* conf) {
    int changed = EXIT_FAILURE;
    const char* names[2] = {GRID_ROWS, GRID_COLS};
    for (int i = 0; i < 2; ++i) {
        const char* s = getenv(names[i]);
        if (!s) continue;
        char* endp;
        unsigned long v = strtoul(s, &endp, 10);
        if (endp == s || v >= USHRT_MAX) fprintf(stderr, "getConfig: bad %s=%s\n", names[i], s);
        else if (i == 0) conf->gridRows = (ushort)v;
        else conf->gridCols = (ushort)v;
        changed = EXIT_SUCCESS;
    }
    return changed;
}

int fillRndVector(ulong size, double* v) {
    if (urndFd < 0 && init_urndfd()) return EXIT_FAILURE;
    for (ulong i = 0; i < size; ++i) {
        double raw, val;
        do {
            if (readFully(urndFd, &raw, sizeof raw)) { ERRPRINT("read_wrap failed to read rnd double\n"); return EXIT_FAILURE; }
            val = sin(raw) * MAXRND;
        } while (!isfinite(val));
        v[i] = val;
    }
    return EXIT_SUCCESS;
}

int doubleVectorsDiff(double* a, double* b, ulong n, double* diffMax) {
    int out = EXIT_SUCCESS;
    double worst = 0;
    ulong reported = 0;
    for (ulong i = 0; i < n; ++i) {
        const double diff = a[i] - b[i];
        const double mag = fabs(diff);
        if (!(mag <= DOUBLE_DIFF_THREASH)) {          /* also true for NaN */
            out = EXIT_FAILURE;
            if (reported++ < 16)
                ERRPRINTS("DOUBLE VECTORS DIFF: DOUBLE_DIFF_THREASH=%lf\t<\t|%13lg| = %lf %% of @a[%lu]\n",
                          DOUBLE_DIFF_THREASH, diff, 100 * mag / fabs(a[i]), i);
#ifdef DOUBLE_VECT_DIFF_EARLY_EXIT
            if (diffMax) *diffMax = diff;
            return EXIT_FAILURE;
#endif
        }
        if (fabs(worst) < mag || isnan(diff)) worst = diff;
    }
    if (reported > 16) ERRPRINTS("... %lu more elements over the threshold\n", reported - 16);
    if (diffMax) *diffMax = worst;
    return out;
}

void statsAvgVar(double* values, uint numVals, double* out) {
    double sum = 0, sumSq = 0;
    for (uint i = 0; i < numVals; ++i) { sum += values[i]; sumSq += values[i] * values[i]; }
    out[0] = sum / numVals;
    out[1] = sumSq / numVals - out[0] * out[0];
}

void printVector(double* v, ulong size) {
    for (ulong i = 0; i < size; ++i) printf("%1.1lf ", v[i]);
    printf("\n");
}

/* ---- compressed inputs ------------------------------------------
 * The reference shells out: system("gzip -d -c <path> > <tmp>") etc. (utils.c:433-462).
 * Here the file is inflated in-process: .gz through zlib, .bz2 through libbz2's
 * high-level API resolved with dlopen (no bzlib.h in the image); .xz / .zip are
 * reported as unsupported.  Returns 0 on success, -1 when `path` has no known
 * compression suffix (caller then opens it as is), 1 on failure. */
static int endsWith(const char* s, const char* suffix) {
    const size_t a = strlen(s), b = strlen(suffix);
    return a >= b && !strcmp(s + a - b, suffix);
}

int extractInTmpFS(char* path, char* tmpFsDecompressPath) {
    const int gz = endsWith(path, ".gz"), bz = endsWith(path, ".bz2");
    if (endsWith(path, ".xz") || endsWith(path, ".zip")) {
        ERRPRINTS("NOT SUPPORTED DECOMPRESS FOR %s (only .gz and .bz2 are inflated in-process)\n", path);
        return 1;
    }
    if (!gz && !bz) return -1;
    FILE* out = fopen(tmpFsDecompressPath, "wb");
    if (!out) { perror("fopen decompress target"); return 1; }
    int rc = 1;
    static char buf[1 << 16];
    if (gz) {
        gzFile in = gzopen(path, "rb");
        if (!in) { perror("gzopen"); goto done; }
        int n;
        while ((n = gzread(in, buf, sizeof buf)) > 0)
            if (fwrite(buf, 1, (size_t)n, out) != (size_t)n) { n = -1; break; }
        rc = n < 0 ? 1 : 0;
        if (gzclose(in) != Z_OK) rc = 1;
    } else {
        void* so = dlopen("libbz2.so.1.0", RTLD_NOW);
        if (!so) so = dlopen("libbz2.so.1", RTLD_NOW);
        if (!so) { ERRPRINTS("cannot load libbz2: %s\n", dlerror()); goto done; }
        void* (*bzopen)(const char*, const char*) = (void* (*)(const char*, const char*))dlsym(so, "BZ2_bzopen");
        int (*bzread)(void*, void*, int) = (int (*)(void*, void*, int))dlsym(so, "BZ2_bzread");
        void (*bzclose)(void*) = (void (*)(void*))dlsym(so, "BZ2_bzclose");
        void* in = bzopen && bzread && bzclose ? bzopen(path, "rb") : NULL;
        if (!in) { ERRPRINTS("BZ2_bzopen failed for %s\n", path); dlclose(so); goto done; }
        int n;
        while ((n = bzread(in, buf, (int)sizeof buf)) > 0)
            if (fwrite(buf, 1, (size_t)n, out) != (size_t)n) { n = -1; break; }
        rc = n < 0 ? 1 : 0;
        bzclose(in);
        dlclose(so);
    }
done:
    if (fclose(out)) rc = 1;
    if (rc) ERRPRINTS("decompression of %s failed\n", path);
    return rc;
}
