/* tests/h5stub/h5stub.c — a stand-in for the corner of libhdf5 that a filter plugin's "set local" callback touches, so that
 * libsz3hip's HDF5 face (sz3_amd/csrc/sz3hip_h5z.cpp) can be driven without HDF5 (not in this image): dataset creation property
 * lists that hold a filter pipeline (id, flags, cd_values), datatypes (class, size, sign), simple dataspaces (rank, extents).
 * Signatures and constants are HDF5's public ones (H5Ppublic.h, H5Tpublic.h, H5Spublic.h, H5Zpublic.h; hid_t = int64_t).
 * Test infrastructure only; built by tests/test_h5z_cpu.py with gcc. The h5stub_* functions make the objects a test needs. */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t hid_t;
typedef int herr_t;
typedef unsigned long long hsize_t;

#define MAX_OBJ 64
#define MAX_FILT 8
struct filt { int id; unsigned flags; size_t n; unsigned cd[64]; };
struct plist { int used, nf; struct filt f[MAX_FILT]; };
struct dtype { int used, cls, sign; size_t size; };
struct space { int used, rank; hsize_t dims[32]; };
static struct plist g_pl[MAX_OBJ];
static struct dtype g_ty[MAX_OBJ];
static struct space g_sp[MAX_OBJ];
int h5stub_calls = 0; /* how many H5* entry points were called (a test checks the plugin really came through here) */

hid_t h5stub_plist_new(void) {
    for (int i = 0; i < MAX_OBJ; i++)
        if (!g_pl[i].used) { memset(&g_pl[i], 0, sizeof(g_pl[i])); g_pl[i].used = 1; return 1000 + i; }
    return -1;
}
hid_t h5stub_type_new(int cls, size_t size, int sign) {
    for (int i = 0; i < MAX_OBJ; i++)
        if (!g_ty[i].used) { g_ty[i].used = 1; g_ty[i].cls = cls; g_ty[i].size = size; g_ty[i].sign = sign; return 2000 + i; }
    return -1;
}
hid_t h5stub_space_new(int rank, const hsize_t *dims) {
    for (int i = 0; i < MAX_OBJ; i++)
        if (!g_sp[i].used) { g_sp[i].used = 1; g_sp[i].rank = rank; memcpy(g_sp[i].dims, dims, rank * sizeof(hsize_t)); return 3000 + i; }
    return -1;
}
void h5stub_reset(void) { memset(g_pl, 0, sizeof(g_pl)); memset(g_ty, 0, sizeof(g_ty)); memset(g_sp, 0, sizeof(g_sp)); h5stub_calls = 0; }
static struct plist *pl(hid_t id) { return id >= 1000 && id < 1000 + MAX_OBJ && g_pl[id - 1000].used ? &g_pl[id - 1000] : 0; }
static struct filt *find(struct plist *p, int filter) {
    for (int i = 0; p && i < p->nf; i++)
        if (p->f[i].id == filter) return &p->f[i];
    return 0;
}

herr_t H5Pset_filter(hid_t plist_id, int filter, unsigned flags, size_t cd_nelmts, const unsigned cd_values[]) {
    h5stub_calls++;
    struct plist *p = pl(plist_id);
    if (!p || p->nf == MAX_FILT || cd_nelmts > 64) return -1;
    struct filt *f = &p->f[p->nf++]; /* (like HDF5: a second H5Pset_filter of the same id APPENDS a second entry) */
    f->id = filter; f->flags = flags; f->n = cd_nelmts;
    if (cd_nelmts) memcpy(f->cd, cd_values, cd_nelmts * sizeof(unsigned));
    return 0;
}
herr_t H5Pmodify_filter(hid_t plist_id, int filter, unsigned flags, size_t cd_nelmts, const unsigned cd_values[]) {
    h5stub_calls++;
    struct filt *f = find(pl(plist_id), filter);
    if (!f || cd_nelmts > 64) return -1;
    f->flags = flags; f->n = cd_nelmts;
    if (cd_nelmts) memcpy(f->cd, cd_values, cd_nelmts * sizeof(unsigned));
    return 0;
}
int H5Pget_nfilters(hid_t plist_id) { h5stub_calls++; struct plist *p = pl(plist_id); return p ? p->nf : -1; }
static void give(const struct filt *f, unsigned *flags, size_t *cd_nelmts, unsigned cd_values[], unsigned *filter_config) {
    if (flags) *flags = f->flags;
    if (cd_nelmts) {
        const size_t room = *cd_nelmts; /* in: capacity of cd_values; out: number of values the filter has */
        if (cd_values) memcpy(cd_values, f->cd, (room < f->n ? room : f->n) * sizeof(unsigned));
        *cd_nelmts = f->n;
    }
    if (filter_config) *filter_config = 3; /* H5Z_FILTER_CONFIG_ENCODE_ENABLED | DECODE_ENABLED */
}
int H5Pget_filter2(hid_t plist_id, unsigned idx, unsigned *flags, size_t *cd_nelmts, unsigned cd_values[], size_t namelen, char name[], unsigned *filter_config) {
    h5stub_calls++;
    struct plist *p = pl(plist_id);
    if (!p || (int)idx >= p->nf) return -1;
    give(&p->f[idx], flags, cd_nelmts, cd_values, filter_config);
    if (namelen && name) name[0] = 0;
    return p->f[idx].id;
}
herr_t H5Pget_filter_by_id2(hid_t plist_id, int filter, unsigned *flags, size_t *cd_nelmts, unsigned cd_values[], size_t namelen, char name[], unsigned *filter_config) {
    h5stub_calls++;
    struct filt *f = find(pl(plist_id), filter);
    if (!f) return -1;
    give(f, flags, cd_nelmts, cd_values, filter_config);
    if (namelen && name) name[0] = 0;
    return 0;
}
int H5Zfilter_avail(int filter) { h5stub_calls++; return filter == 32024; }
int H5Tget_class(hid_t t) { h5stub_calls++; return t >= 2000 && t < 2000 + MAX_OBJ && g_ty[t - 2000].used ? g_ty[t - 2000].cls : -1; }
size_t H5Tget_size(hid_t t) { h5stub_calls++; return t >= 2000 && t < 2000 + MAX_OBJ && g_ty[t - 2000].used ? g_ty[t - 2000].size : 0; }
int H5Tget_sign(hid_t t) { h5stub_calls++; return t >= 2000 && t < 2000 + MAX_OBJ && g_ty[t - 2000].used ? g_ty[t - 2000].sign : -1; }
int H5Sget_simple_extent_dims(hid_t s, hsize_t dims[], hsize_t maxdims[]) {
    h5stub_calls++;
    if (s < 3000 || s >= 3000 + MAX_OBJ || !g_sp[s - 3000].used) return -1;
    if (dims) memcpy(dims, g_sp[s - 3000].dims, g_sp[s - 3000].rank * sizeof(hsize_t));
    if (maxdims) memcpy(maxdims, g_sp[s - 3000].dims, g_sp[s - 3000].rank * sizeof(hsize_t));
    return g_sp[s - 3000].rank;
}
