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
        const size_t want = cap - n, got = fread(out + n, sizeof *out, want, fp);
        n += got;
        if (got < want) break;                       /* end of file or error: no further read on this stream */
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

/* whole file -> malloc'ed buffer (caller frees); NULL on error */
static unsigned char* slurp(const char* path, size_t* len) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror("fopen"); return NULL; }
    unsigned char* b = NULL;
    if (!fseek(f, 0, SEEK_END)) {
        const long n = ftell(f);
        if (n >= 0 && !fseek(f, 0, SEEK_SET) && (b = malloc((size_t)n + 1)) && fread(b, 1, (size_t)n, f) == (size_t)n) *len = (size_t)n;
        else { free(b); b = NULL; }
    }
    fclose(f);
    return b;
}

/* .xz: liblzma's single-call buffer decoder, loaded at run time (the image has the library but not its headers).
 * The size of the output comes from the file itself: the stream footer (last 12 bytes: CRC32, backward size, flags,
 * "YZ") points at the index, whose records hold the uncompressed size of every block (xz file format 1.x, sections 2.1.2
 * and 4).  A truncated file has no footer and is refused at once -- the decoder's BUF_ERROR does not tell a short input
 * from a short output buffer, so growing the buffer until it stops complaining never ends for such a file.  One stream
 * per file, which is what xz and every library writer produce. */
static int xzVarint(const unsigned char* p, const unsigned char* end, uint64_t* out, const unsigned char** next) {
    uint64_t v = 0;
    for (int k = 0; k < 9 && p < end; ++k, ++p) {
        v |= (uint64_t)(*p & 0x7F) << (7 * k);
        if (!(*p & 0x80)) { *out = v; *next = p + 1; return 1; }
    }
    return 0;
}
static int xzUncompressedSize(const unsigned char* in, size_t n, uint64_t* total) {
    while (n >= 4 && !in[n - 1] && !in[n - 2] && !in[n - 3] && !in[n - 4]) n -= 4;      /* stream padding */
    if (n < 12 + 12 || in[n - 2] != 'Y' || in[n - 1] != 'Z') return 0;
    const uint64_t indexSize = ((uint64_t)(in[n - 8] | in[n - 7] << 8 | in[n - 6] << 16 | (uint64_t)in[n - 5] << 24) + 1) * 4;
    if (indexSize + 12 + 12 > n) return 0;
    const unsigned char *p = in + n - 12 - indexSize, *end = in + n - 12;
    uint64_t records, unpadded, size;
    if (*p++ != 0 || !xzVarint(p, end, &records, &p) || records > indexSize) return 0;
    *total = 0;
    for (uint64_t k = 0; k < records; ++k) {
        if (!xzVarint(p, end, &unpadded, &p) || !xzVarint(p, end, &size, &p) || size > ((uint64_t)1 << 40)) return 0;
        *total += size;
    }
    return 1;
}
static int inflateXz(const char* path, FILE* out) {
    void* so = dlopen("liblzma.so.5", RTLD_NOW);
    if (!so) { ERRPRINTS("cannot load liblzma: %s\n", dlerror()); return 1; }
    typedef int (*decode_fn)(uint64_t*, uint32_t, const void*, const uint8_t*, size_t*, size_t, uint8_t*, size_t*, size_t);
    decode_fn decode = (decode_fn)dlsym(so, "lzma_stream_buffer_decode");
    size_t inLen = 0;
    unsigned char* in = decode ? slurp(path, &inLen) : NULL;
    int rc = 1;
    if (in) {
        uint64_t total = 0;
        unsigned char* o = NULL;
        if (!xzUncompressedSize(in, inLen, &total)) ERRPRINTS("%s: no xz stream footer / index (truncated file?)\n", path);
        else if (total > ((uint64_t)1 << 40) || !(o = malloc(total ? (size_t)total : 1))) ERRPRINTS("%s: %lu bytes to inflate\n", path, (ulong)total);
        else {
            uint64_t memlimit = UINT64_MAX;
            size_t inPos = 0, outPos = 0;
            const int r = decode(&memlimit, 0, NULL, in, &inPos, inLen, o, &outPos, (size_t)total);      /* LZMA_OK = 0 */
            if (r == 0) rc = fwrite(o, 1, outPos, out) == outPos ? 0 : 1;
        }
        free(o);
        free(in);
    }
    dlclose(so);
    return rc;
}

/* .zip: the FIRST member of the archive (the reference unzips and reads the one file inside), method 0 (stored) or
 * 8 (deflate, raw stream through zlib).  Sizes come from the central directory (members written by streaming zippers
 * have zeros in their local header). */
static uint32_t le32(const unsigned char* p) { return p[0] | p[1] << 8 | p[2] << 16 | (uint32_t)p[3] << 24; }
static uint16_t le16(const unsigned char* p) { return (uint16_t)(p[0] | p[1] << 8); }
static int inflateZip(const char* path, FILE* out) {
    size_t n = 0;
    unsigned char* z = slurp(path, &n);
    if (!z) return 1;
    int rc = 1;
    size_t eocd = n;                                 /* end-of-central-directory record: PK\5\6, within the last 64 KiB */
    for (size_t i = n >= 22 ? n - 22 : 0; n >= 22 && i + 65557 >= n; --i) {
        if (le32(z + i) == 0x06054b50u) { eocd = i; break; }
        if (i == 0) break;
    }
    if (eocd == n) { ERRPRINTS("%s: no zip end-of-central-directory record\n", path); goto done; }
    {
        const size_t cd = le32(z + eocd + 16);
        if (cd + 46 > n || le32(z + cd) != 0x02014b50u) { ERRPRINTS("%s: bad zip central directory\n", path); goto done; }
        const uint16_t method = le16(z + cd + 10);
        const size_t csize = le32(z + cd + 20), usize = le32(z + cd + 24), lh = le32(z + cd + 42);
        if (lh + 30 > n || le32(z + lh) != 0x04034b50u) { ERRPRINTS("%s: bad zip local header\n", path); goto done; }
        const size_t data = lh + 30 + le16(z + lh + 26) + le16(z + lh + 28);
        if (csize == 0xFFFFFFFFu || usize == 0xFFFFFFFFu || data + csize > n) { ERRPRINTS("%s: zip64 / truncated member not supported\n", path); goto done; }
        if (method == 0) rc = fwrite(z + data, 1, csize, out) == csize ? 0 : 1;
        else if (method == 8) {
            unsigned char* o = malloc(usize ? usize : 1);
            z_stream st;
            memset(&st, 0, sizeof st);
            if (o && inflateInit2(&st, -MAX_WBITS) == Z_OK) {
                st.next_in = z + data; st.avail_in = (uInt)csize; st.next_out = o; st.avail_out = (uInt)usize;
                const int r = inflate(&st, Z_FINISH);
                if (r == Z_STREAM_END && st.total_out == usize) rc = fwrite(o, 1, usize, out) == usize ? 0 : 1;
                inflateEnd(&st);
            }
            free(o);
        } else ERRPRINTS("%s: zip compression method %u not supported\n", path, method);
    }
done:
    free(z);
    return rc;
}

int extractInTmpFS(char* path, char* tmpFsDecompressPath) {
    const int gz = endsWith(path, ".gz"), bz = endsWith(path, ".bz2");
    if (endsWith(path, ".xz") || endsWith(path, ".zip")) {
        FILE* o = fopen(tmpFsDecompressPath, "wb");
        if (!o) { perror("fopen decompress target"); return 1; }
        int r = endsWith(path, ".xz") ? inflateXz(path, o) : inflateZip(path, o);
        if (fclose(o)) r = 1;
        if (r) ERRPRINTS("decompression of %s failed\n", path);
        return r;
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
