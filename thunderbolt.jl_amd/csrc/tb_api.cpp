// tb_api.cpp — C-ABI entry points (include/tbhip.h): objects, memory, argument checking, error strings.
// No arithmetic of the path lives here; kernels are in tb_assembly.hip / tb_reaction.hip / tb_algebra.hip.
#include <algorithm>
#include <mutex>
#include <cmath>
#include <cstdarg>
#include <cstring>

#include <omp.h>

#include "tb_internal.h"

namespace tb {

static thread_local std::string g_error;

void set_error(const char *fmt, ...)
{
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_error = buf;
}

// name of the element / assembly kernel instance the latest assembly call of this thread launched (tb_last_kernel_name): what a benchmark line
// reports as "the kernel that ran" instead of a name typed into the script
static thread_local std::string g_last_kernel;
void set_last_kernel(const char *fmt, ...)
{
    char buf[256];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_kernel = buf;
}
const char *last_kernel() { return g_last_kernel.c_str(); }

int reset_status(tb_device *dev)
{
    if (dev->defer_status) return TB_OK; // flags stay raised (they are only ever set) until tb_device_poll_status reads and clears them
    TB_HIP(hipMemsetAsync(dev->d_status, 0, sizeof(Status), dev->stream));
    return TB_OK;
}

static int read_status(tb_device *dev);
int check_status(tb_device *dev)
{
    if (dev->defer_status) return TB_OK;
    return read_status(dev);
}

static int read_status(tb_device *dev)
{
    TB_NO_CAPTURE(dev);
    TB_HIP(hipMemcpyAsync(dev->h_status, dev->d_status, sizeof(Status), hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    if (dev->h_status->neg_detj) {
        set_error("detJ <= 0 in cell %lld (0-based)", dev->h_status->cell);
        return TB_ERR_NEG_DETJ;
    }
    if (dev->h_status->pattern_missing) {
        set_error("coupling of cell %lld (0-based) missing from the CSR pattern", dev->h_status->cell);
        return TB_ERR_PATTERN;
    }
    return TB_OK;
}

int read_status_public(tb_device *dev) { return read_status(dev); }

template <class T>
static void copy_rebased(std::vector<T> &dst, const T *src, size_t n, int base)
{
    dst.resize(n);
    for (size_t i = 0; i < n; ++i) dst[i] = src[i] - (T)base;
}

} // namespace tb

using namespace tb;

extern "C" {

const char *tb_last_error_string(void) { return g_error.c_str(); }
const char *tb_version(void) { return "thunderbolt.jl_amd 0.1 (gfx950)"; }
int tb_abi_revision(void) { return TB_ABI_REVISION; }

// Host threads for the plan builders (OpenMP loops of tb_plans.cpp / tb_hostgen.cpp).  Inside a container the CPU set usually still lists every
// hardware thread of the machine while a CFS quota caps the CPU TIME (cgroup v2 cpu.max "quota period", v1 cpu.cfs_quota_us / cpu.cfs_period_us): 256
// OpenMP threads under a 16-CPU quota are throttled in lock-step — the 216³ plan build measured 21 s that way.  Applied once, at the first device:
// the default team size becomes min(what OpenMP would take, ⌈quota⌉); an explicit OMP_NUM_THREADS is left alone.
static int cgroup_cpu_quota()
{
    double q = 0.0;
    if (FILE *f = fopen("/sys/fs/cgroup/cpu.max", "r")) {
        char a[64] = ""; double per = 0.0;
        if (fscanf(f, "%63s %lf", a, &per) == 2 && strcmp(a, "max") != 0 && per > 0) q = atof(a) / per;
        fclose(f);
    } else if (FILE *g = fopen("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "r")) {
        double quota = -1, per = 0;
        if (fscanf(g, "%lf", &quota) != 1) quota = -1;
        fclose(g);
        if (FILE *h = fopen("/sys/fs/cgroup/cpu/cpu.cfs_period_us", "r")) { if (fscanf(h, "%lf", &per) != 1) per = 0; fclose(h); }
        if (quota > 0 && per > 0) q = quota / per;
    }
    return q > 0 ? std::max(1, (int)std::ceil(q - 1e-9)) : 0;
}
static void cap_host_threads_once()
{
    static const bool done = [] {
        if (getenv("OMP_NUM_THREADS")) return true;
        const int q = cgroup_cpu_quota();
        if (q > 0 && q < omp_get_max_threads()) omp_set_num_threads(q);
        return true;
    }();
    (void)done;
}
const char *tb_last_kernel_name(void) { return tb::last_kernel(); }

// ------------------------------------------------------------------ device
int tb_device_create(int hip_device_id, tb_device **out)
{
    TB_REQUIRE(out, "tb_device_create: out is NULL");
    *out = nullptr;
    int n = 0;
    TB_HIP(hipGetDeviceCount(&n));
    TB_REQUIRE(hip_device_id >= 0 && hip_device_id < n, "tb_device_create: device %d of %d", hip_device_id, n);
    TB_HIP(hipSetDevice(hip_device_id));
    cap_host_threads_once();
    std::unique_ptr<tb_device, int (*)(tb_device *)> dev(new tb_device, tb_device_destroy); // a failure below releases what was created so far
    dev->id = hip_device_id;
    hipDeviceProp_t prop;
    TB_HIP(hipGetDeviceProperties(&prop, hip_device_id));
    dev->n_cu = prop.multiProcessorCount;
    dev->hbm_bytes = prop.totalGlobalMem;
    dev->name = prop.gcnArchName;
    TB_HIP(hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking));
    TB_HIP(hipMalloc((void **)&dev->d_status, sizeof(Status)));
    TB_HIP(hipMalloc((void **)&dev->d_slots, 8 * 1024 * sizeof(double)));
    TB_HIP(hipMemset(dev->d_slots, 0, 8 * 1024 * sizeof(double)));
    TB_HIP(hipHostMalloc((void **)&dev->h_status, sizeof(Status), hipHostMallocDefault));
    *out = dev.release();
    return TB_OK;
}

int tb_device_destroy(tb_device *dev)
{
    if (!dev) return TB_OK;
    hipSetDevice(dev->id);
    if (dev->own_stream && dev->stream) hipStreamDestroy(dev->stream);
    if (dev->aux_stream) hipStreamDestroy(dev->aux_stream);
    for (hipEvent_t e : dev->aux_ev) if (e) hipEventDestroy(e);
    if (dev->d_status) hipFree(dev->d_status);
    if (dev->h_status) hipHostFree(dev->h_status);
    if (dev->d_scratch) hipFree(dev->d_scratch);
    if (dev->d_tslot) hipFree(dev->d_tslot);
    if (dev->d_slots) hipFree(dev->d_slots);
    delete dev;
    return TB_OK;
}

int tb_device_set_stream(tb_device *dev, void *hip_stream)
{
    TB_REQUIRE(dev, "tb_device_set_stream: dev is NULL");
    TB_SYNC_STREAM(dev);
    if (dev->own_stream && dev->stream) TB_HIP(hipStreamDestroy(dev->stream));
    if (hip_stream) {
        dev->stream = (hipStream_t)hip_stream;
        dev->own_stream = false;
    } else {
        TB_HIP(hipStreamCreateWithFlags(&dev->stream, hipStreamNonBlocking));
        dev->own_stream = true;
    }
    return TB_OK;
}

int tb_device_use_null_stream(tb_device *dev)
{
    TB_REQUIRE(dev, "tb_device_use_null_stream: dev is NULL");
    TB_SYNC_STREAM(dev);
    if (dev->own_stream && dev->stream) TB_HIP(hipStreamDestroy(dev->stream));
    dev->stream = nullptr; // the legacy default stream: ordered with every blocking stream of the process (a host framework's default stream is this one)
    dev->own_stream = false;
    return TB_OK;
}

int tb_device_defer_status(tb_device *dev, int on)
{
    TB_REQUIRE(dev, "tb_device_defer_status: NULL device");
    TB_HIP(hipSetDevice(dev->id));
    if (on && !dev->defer_status) TB_HIP(hipMemsetAsync(dev->d_status, 0, sizeof(Status), dev->stream)); // start from a clean block
    dev->defer_status = on != 0;
    return TB_OK;
}

int tb_device_poll_status(tb_device *dev)
{
    TB_REQUIRE(dev, "tb_device_poll_status: NULL device");
    TB_NO_CAPTURE(dev); // (before anything is enqueued: the status reset below would otherwise become a node of the capture)
    TB_HIP(hipSetDevice(dev->id));
    const int rc = tb::read_status_public(dev);
    TB_HIP(hipMemsetAsync(dev->d_status, 0, sizeof(Status), dev->stream));
    return rc;
}

int tb_device_synchronize(tb_device *dev)
{
    TB_REQUIRE(dev, "tb_device_synchronize: dev is NULL");
    TB_SYNC_STREAM(dev);
    return TB_OK;
}

int tb_device_info(tb_device *dev, char *name, size_t name_len, int *n_cu, size_t *hbm_bytes)
{
    TB_REQUIRE(dev, "tb_device_info: dev is NULL");
    if (name && name_len) { strncpy(name, dev->name.c_str(), name_len - 1); name[name_len - 1] = 0; }
    if (n_cu) *n_cu = dev->n_cu;
    if (hbm_bytes) *hbm_bytes = dev->hbm_bytes;
    return TB_OK;
}

int tb_malloc(tb_device *dev, size_t bytes, void **d_ptr)
{
    TB_REQUIRE(dev && d_ptr, "tb_malloc: NULL argument");
    *d_ptr = nullptr;
    TB_HIP(hipSetDevice(dev->id));
    if (bytes == 0) return TB_OK;
    hipError_t e = hipMalloc(d_ptr, bytes);
    if (e != hipSuccess) { set_error("tb_malloc(%zu): %s", bytes, hipGetErrorString(e)); return TB_ERR_NOMEM; }
    return TB_OK;
}

// an array rewritten through one of the raw-array entries loses its sliced mirror.  The patterns that hold mirrors are listed process-wide (a pattern
// leaves the list in tb_pattern_destroy without touching its mesh or device, which a host's finaliser order may already have released)
static std::vector<tb_pattern *> &g_mirrored = *new std::vector<tb_pattern *>; // (never destroyed: a finaliser may still call tb_pattern_destroy while the process exits)
static std::mutex &g_mirrored_mutex = *new std::mutex;
// bytes: extent of the write [d_dst, d_dst + bytes) — any overlap with a bound array invalidates its mirror (a memset / copy over a larger arena that
// holds the array as a sub-range starts BEFORE it).  bytes = 0: extent unknown (tb_free of an allocation whose size the library does not track) — the
// binding goes when the bound array starts at or behind d_dst … which for a free can only be the allocation itself or a sub-range the host carved from it;
// sub-ranges of a freed arena whose start lies further in are the host's to unbind (tbhip.h: "an array is unbound before it is freed").
static void mirror_drop_any(tb_device *, const void *d_dst, size_t bytes = 0)
{
    std::lock_guard<std::mutex> lock(g_mirrored_mutex);
    const char *w0 = (const char *)d_dst, *w1 = w0 + (bytes ? bytes : 1);
    for (tb_pattern *p : g_mirrored)
        for (const double *&q : p->mir_nz) {
            if (!q) continue;
            const char *a0 = (const char *)q, *a1 = (const char *)(q + p->nnz);
            if (w0 < a1 && w1 > a0) q = nullptr;
        }
}

int tb_free(tb_device *dev, void *d_ptr)
{
    TB_REQUIRE(dev, "tb_free: dev is NULL");
    if (d_ptr) {
        {   // every binding whose array starts inside the freed allocation goes (hipMemGetAddressRange: the allocation's own extent)
            void *base = nullptr; size_t sz = 0;
            if (hipMemGetAddressRange((hipDeviceptr_t *)&base, &sz, (hipDeviceptr_t)d_ptr) == hipSuccess && base == d_ptr && sz) mirror_drop_any(dev, d_ptr, sz);
            else { (void)hipGetLastError(); mirror_drop_any(dev, d_ptr); }
        }
        // a binding is the address: the allocator may hand it to the next matrix (host finalisers free without unmirror!)
        TB_HIP(hipFree(d_ptr));
    }
    return TB_OK;
}

int tb_memcpy_h2d(tb_device *dev, void *d_dst, const void *src, size_t bytes)
{
    TB_REQUIRE(dev && (bytes == 0 || (d_dst && src)), "tb_memcpy_h2d: NULL argument");
    if (!bytes) return TB_OK;
    TB_NO_CAPTURE(dev); // the copy waits for the stream (src is the caller's to reuse on return)
    mirror_drop_any(dev, d_dst, bytes);
    TB_HIP(hipMemcpyAsync(d_dst, src, bytes, hipMemcpyHostToDevice, dev->stream));
    TB_SYNC_STREAM(dev);
    return TB_OK;
}

int tb_memcpy_d2h(tb_device *dev, void *dst, const void *d_src, size_t bytes)
{
    TB_REQUIRE(dev && (bytes == 0 || (dst && d_src)), "tb_memcpy_d2h: NULL argument");
    if (!bytes) return TB_OK;
    TB_NO_CAPTURE(dev);
    TB_HIP(hipMemcpyAsync(dst, d_src, bytes, hipMemcpyDeviceToHost, dev->stream));
    TB_SYNC_STREAM(dev);
    return TB_OK;
}

int tb_memcpy_d2d(tb_device *dev, void *d_dst, const void *d_src, size_t bytes)
{
    TB_REQUIRE(dev && (bytes == 0 || (d_dst && d_src)), "tb_memcpy_d2d: NULL argument");
    if (!bytes) return TB_OK;
    mirror_drop_any(dev, d_dst, bytes); // copyto!(A.nzval, …) of the Julia host comes through here
    TB_HIP(hipMemcpyAsync(d_dst, d_src, bytes, hipMemcpyDeviceToDevice, dev->stream));
    return TB_OK;
}

int tb_memset(tb_device *dev, void *d_ptr, int byte, size_t bytes)
{
    TB_REQUIRE(dev && (bytes == 0 || d_ptr), "tb_memset: NULL argument");
    if (!bytes) return TB_OK;
    mirror_drop_any(dev, d_ptr, bytes);
    TB_HIP(hipMemsetAsync(d_ptr, byte, bytes, dev->stream));
    return TB_OK;
}

int tb_event_create(tb_device *dev, void **event)
{
    TB_REQUIRE(dev && event, "tb_event_create: NULL argument");
    hipEvent_t e;
    TB_HIP(hipEventCreate(&e));
    *event = (void *)e;
    return TB_OK;
}
int tb_event_record(tb_device *dev, void *event)
{
    TB_REQUIRE(dev && event, "tb_event_record: NULL argument");
    TB_HIP(hipEventRecord((hipEvent_t)event, dev->stream));
    return TB_OK;
}
int tb_event_elapsed_ms(void *start, void *stop, float *ms)
{
    TB_REQUIRE(start && stop && ms, "tb_event_elapsed_ms: NULL argument");
    TB_HIP(hipEventSynchronize((hipEvent_t)stop));
    TB_HIP(hipEventElapsedTime(ms, (hipEvent_t)start, (hipEvent_t)stop));
    return TB_OK;
}
int tb_event_destroy(void *event)
{
    if (event) TB_HIP(hipEventDestroy((hipEvent_t)event));
    return TB_OK;
}

// ------------------------------------------------------------------ mesh
int tb_mesh_create(tb_device *dev, int geom_kind, int64_t n_nodes, const double *xyz, int64_t n_cells,
                   const int32_t *conn, int field_kind, int ncomp, const int32_t *cell_dofs, int64_t ndofs,
                   int index_base, tb_mesh **out)
{
    TB_REQUIRE(dev && out && xyz && conn && cell_dofs, "tb_mesh_create: NULL argument");
    *out = nullptr;
    TB_REQUIRE(geom_kind == TB_HEX8 || geom_kind == TB_TET4 || geom_kind == TB_QUAD4, "tb_mesh_create: geometry kind %d (need TB_QUAD4, TB_HEX8 or TB_TET4)", geom_kind);
    TB_REQUIRE(field_kind == geom_kind || (geom_kind == TB_HEX8 && field_kind == TB_HEX27),
               "tb_mesh_create: field kind %d does not live on geometry kind %d", field_kind, geom_kind);
    TB_REQUIRE(ncomp == 1 || ncomp == 3, "tb_mesh_create: ncomp must be 1 or 3 (got %d)", ncomp);
    TB_REQUIRE(index_base == 0 || index_base == 1, "tb_mesh_create: index_base must be 0 or 1");
    TB_REQUIRE(n_nodes > 0 && n_cells >= 0 && ndofs > 0 && n_nodes < 0x7fffffff && ndofs < 0x7fffffff,
               "tb_mesh_create: sizes out of range (nodes %lld, cells %lld, dofs %lld)", (long long)n_nodes, (long long)n_cells, (long long)ndofs);
    std::unique_ptr<tb_mesh, int (*)(tb_mesh *)> m(new tb_mesh, tb_mesh_destroy); // a failed upload releases the earlier device arrays
    m->dev = dev;
    m->geom_kind = geom_kind; m->field_kind = field_kind; m->ncomp = ncomp;
    m->nverts = kind_nverts(geom_kind); m->nb = kind_nbasis(field_kind); m->ndpc = m->nb * ncomp;
    m->n_nodes = n_nodes; m->n_cells = n_cells; m->ndofs = ndofs;
    m->h_xyz.assign(xyz, xyz + 3 * n_nodes);
    copy_rebased(m->h_conn, conn, (size_t)n_cells * m->nverts, index_base);
    copy_rebased(m->h_cell_dofs, cell_dofs, (size_t)n_cells * m->ndpc, index_base);
    for (int32_t v : m->h_conn) TB_REQUIRE(v >= 0 && v < n_nodes, "tb_mesh_create: node id %d out of range", v + index_base);
    for (int32_t v : m->h_cell_dofs) TB_REQUIRE(v >= 0 && v < ndofs, "tb_mesh_create: dof id %d out of range", v + index_base);
    TB_HIP(hipSetDevice(dev->id));
    int rc;
    if ((rc = upload(dev, m->h_xyz, &m->d_xyz))) return rc;
    if ((rc = upload(dev, m->h_conn, &m->d_conn))) return rc;
    if ((rc = upload(dev, m->h_cell_dofs, &m->d_cell_dofs))) return rc;
    *out = m.release();
    return TB_OK;
}

int tb_mesh_destroy(tb_mesh *m)
{
    if (!m) return TB_OK;
    hipFree(m->d_xyz); hipFree(m->d_conn); hipFree(m->d_cell_dofs); hipFree(m->d_node_dof0); hipFree(m->d_cell_xyz); hipFree(m->d_rank27);
    if (m->colors) hipFree(m->colors->d_cells);
    if (m->ea) { hipFree(m->ea->d_ptr); hipFree(m->ea->d_src); hipFree(m->ea->d_ea); hipFree(m->ea->d_ell); hipFree(m->ea->d_ell_t); }
    free_patch_plan(m);
    free_vec_patch_plans(m);
    delete m;
    return TB_OK;
}

int64_t tb_mesh_ncells(const tb_mesh *m) { return m ? m->n_cells : -1; }
int64_t tb_mesh_ndofs(const tb_mesh *m) { return m ? m->ndofs : -1; }

// ------------------------------------------------------------------ pattern
int tb_pattern_create(tb_mesh *mesh, int64_t n_rows, const int64_t *rowptr, const int32_t *colidx, int index_base,
                      tb_pattern **out)
{
    TB_REQUIRE(mesh && rowptr && colidx && out, "tb_pattern_create: NULL argument");
    *out = nullptr;
    TB_REQUIRE(n_rows == mesh->ndofs, "tb_pattern_create: %lld rows but the dof table has %lld dofs", (long long)n_rows, (long long)mesh->ndofs);
    TB_REQUIRE(index_base == 0 || index_base == 1, "tb_pattern_create: index_base must be 0 or 1");
    std::unique_ptr<tb_pattern, int (*)(tb_pattern *)> p(new tb_pattern, tb_pattern_destroy); // error returns free the device arrays too
    p->mesh = mesh;
    p->n_rows = n_rows;
    copy_rebased(p->h_rowptr, rowptr, (size_t)n_rows + 1, index_base);
    TB_REQUIRE(p->h_rowptr[0] == 0, "tb_pattern_create: rowptr does not start at index_base");
    p->nnz = p->h_rowptr[n_rows];
    for (int64_t r = 0; r < n_rows; ++r) TB_REQUIRE(p->h_rowptr[r + 1] >= p->h_rowptr[r], "tb_pattern_create: rowptr not monotone at row %lld", (long long)r);
    copy_rebased(p->h_colidx, colidx, (size_t)p->nnz, index_base);
    for (int64_t r = 0; r < n_rows; ++r)
        for (int64_t k = p->h_rowptr[r]; k < p->h_rowptr[r + 1]; ++k) {
            TB_REQUIRE(p->h_colidx[k] >= 0 && p->h_colidx[k] < n_rows, "tb_pattern_create: column out of range in row %lld", (long long)r);
            TB_REQUIRE(k == p->h_rowptr[r] || p->h_colidx[k] > p->h_colidx[k - 1], "tb_pattern_create: columns of row %lld not sorted", (long long)r);
        }
    // every cell coupling must exist in the pattern (host check, before anything is allocated on the device; the scatter maps are built lazily)
    {
        const int ndpc = mesh->ndpc;
        bool missing = false;
#pragma omp parallel for schedule(static) reduction(|| : missing)
        for (int64_t c = 0; c < mesh->n_cells; ++c) {
            const int32_t *d = &mesh->h_cell_dofs[c * ndpc];
            for (int i = 0; i < ndpc; ++i) {
                const int32_t *b = &p->h_colidx[p->h_rowptr[d[i]]], *e = &p->h_colidx[p->h_rowptr[d[i] + 1]];
                for (int j = 0; j < ndpc; ++j) {
                    const int32_t *it = std::lower_bound(b, e, d[j]);
                    if (it == e || *it != d[j]) missing = true;
                }
            }
        }
        if (missing) { set_error("tb_pattern_create: a cell coupling is missing from the CSR pattern"); return TB_ERR_PATTERN; }
    }
    TB_HIP(hipSetDevice(mesh->dev->id));
    int rc;
    if ((rc = upload(mesh->dev, p->h_rowptr, &p->d_rowptr))) return rc;
    if ((rc = upload(mesh->dev, p->h_colidx, &p->d_colidx))) return rc;
    *out = p.release();
    return TB_OK;
}

int tb_pattern_destroy(tb_pattern *p)
{
    if (!p) return TB_OK;
    { std::lock_guard<std::mutex> lock(g_mirrored_mutex); g_mirrored.erase(std::remove(g_mirrored.begin(), g_mirrored.end(), p), g_mirrored.end()); }
    hipFree(p->d_rowptr); hipFree(p->d_colidx); hipFree(p->d_blkrow); hipFree(p->d_blkrec); hipFree(p->d_rowsig); hipFree(p->d_sigoff); hipFree(p->d_wrunrec); for (double *q : p->d_mir) hipFree(q); hipFree(p->d_mir_base); hipFree(p->d_mir_off); hipFree(p->d_bcol); hipFree(p->d_diagpos); hipFree(p->d_emap); hipFree(p->d_blockpos); hipFree(p->d_cg_ws); hipFree(p->d_gmres_ws); hipFree(p->d_pcg_ws); hipFree(p->d_cheb_ws); hipFree(p->d_q2pos); hipFree(p->d_q2pos_t); hipFree(p->d_kebuf); hipFree(p->d_qpbuf); hipFree(p->d_gnodes);
    free_patch_mat_plan(p);
    free_patch_fused_plan(p);
    delete p;
    return TB_OK;
}

int64_t tb_pattern_nnz(const tb_pattern *p) { return p ? p->nnz : -1; }
const int64_t *tb_pattern_rowptr_device(const tb_pattern *p) { return p ? p->d_rowptr : nullptr; }
const int32_t *tb_pattern_colidx_device(const tb_pattern *p) { return p ? p->d_colidx : nullptr; }

// ------------------------------------------------------------------ forms
int tb_form_create(tb_mesh *mesh, int form_kind, int qorder, const tb_coef *coef, tb_form **out)
{
    TB_REQUIRE(mesh && coef && out, "tb_form_create: NULL argument");
    *out = nullptr;
    TB_REQUIRE(form_kind == TB_FORM_MASS || form_kind == TB_FORM_DIFFUSION || form_kind == TB_FORM_SOURCE,
               "tb_form_create: unknown form kind %d", form_kind);
    if (qorder == 0) qorder = std::max(2 * kind_order(mesh->field_kind) - 1, 2); // src/discretization/fem.jl:52-55
    auto f = std::make_unique<tb_form>();
    f->mesh = mesh; f->kind = form_kind; f->qorder = qorder; f->coef = *coef; f->coef.field = nullptr;
    // field coefficients are first-order nodal data per cell (coefficients.jl:85-99): the cell's own basis for first-order fields, the
    // eight geometry nodes for the quadratic field
    const int nb = mesh->field_kind == TB_HEX27 ? 8 : mesh->nb;
    int64_t need_field = 0;
    if (form_kind == TB_FORM_MASS) {
        TB_REQUIRE(coef->kind == TB_COEF_CONST_SCALAR || coef->kind == TB_COEF_FIELD_SCALAR, "mass form: coefficient kind %d", coef->kind);
        if (coef->kind == TB_COEF_FIELD_SCALAR) { f->field = true; need_field = mesh->n_cells * nb; }
    } else if (form_kind == TB_FORM_DIFFUSION) {
        double k[9] = {0};
        switch (coef->kind) {
        case TB_COEF_CONST_SCALAR: k[0] = k[4] = k[8] = coef->p[0]; break;
        case TB_COEF_CONST_TENSOR: for (int i = 0; i < 9; ++i) k[i] = coef->p[i]; break;
        case TB_COEF_SPECTRAL_CONST: // λ₁ f⊗f + λ₂ s⊗s + λ₃ n⊗n, microstructure.jl:136-138
            for (int v = 0; v < 3; ++v)
                for (int i = 0; i < 3; ++i)
                    for (int j = 0; j < 3; ++j) k[3 * i + j] += coef->p[9 + v] * coef->p[3 * v + i] * coef->p[3 * v + j];
            break;
        case TB_COEF_TRANSVERSE_CONST: // λ₁ f⊗f + λ₂ (I − f⊗f), microstructure.jl:89-92
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    const double ff = coef->p[i] * coef->p[j];
                    k[3 * i + j] = coef->p[3] * ff + coef->p[4] * ((i == j ? 1.0 : 0.0) - ff);
                }
            break;
        case TB_COEF_SPECTRAL_FIELD: f->field = true; need_field = mesh->n_cells * nb * 9; break;
        case TB_COEF_FIELD_SCALAR: f->field = true; need_field = mesh->n_cells * nb; break; // κ(x)·I, FieldCoefficient (coefficients.jl:85-99)
        default: set_error("diffusion form: coefficient kind %d", coef->kind); return TB_ERR_BAD_ARG;
        }
        const double den = coef->Cm * coef->chi;
        for (int i = 0; i < 9; ++i) f->Dconst[i] = coef->wrap ? k[i] / den : k[i]; // κ/(Cₘ·χ), coefficients.jl:158-161
        f->symmetric = f->Dconst[1] == f->Dconst[3] && f->Dconst[2] == f->Dconst[6] && f->Dconst[5] == f->Dconst[7];
    } else {
        TB_REQUIRE(coef->kind >= TB_SRC_CONST && coef->kind <= TB_SRC_TABULATED, "source form: source kind %d", coef->kind);
    }
    if (need_field) {
        TB_REQUIRE(coef->field && coef->field_len == need_field, "tb_form_create: field coefficient needs %lld values, got %lld",
                   (long long)need_field, (long long)coef->field_len);
        TB_HIP(hipMalloc((void **)&f->d_field, sizeof(double) * need_field));
        TB_HIP(hipMemcpyAsync(f->d_field, coef->field, sizeof(double) * need_field, hipMemcpyHostToDevice, mesh->dev->stream));
        TB_SYNC_STREAM(mesh->dev);
    }
    *out = f.release();
    return TB_OK;
}

int tb_form_destroy(tb_form *f)
{
    if (!f) return TB_OK;
    hipFree(f->d_field); hipFree(f->d_dtab); hipFree(f->d_gtab); hipFree(f->d_table); hipFree(f->d_facets); hipFree(f->d_act_field); hipFree(f->d_qp_buf); hipFree(f->d_cellset); if (f->set_colors) hipFree(f->set_colors->d_cells);
    delete f;
    return TB_OK;
}

int tb_form_set_table(tb_form *f, const double *values, int64_t n)
{
    TB_REQUIRE(f && values, "tb_form_set_table: NULL argument");
    TB_REQUIRE(f->kind == TB_FORM_SOURCE && f->coef.kind == TB_SRC_TABULATED, "tb_form_set_table: form is not a tabulated source");
    const int nq = f->mesh->field_kind == TB_TET4 ? (f->qorder == 1 ? 1 : 4) : f->qorder * f->qorder * f->qorder;
    TB_REQUIRE(n == f->mesh->n_cells * nq, "tb_form_set_table: expected %lld values, got %lld", (long long)(f->mesh->n_cells * nq), (long long)n);
    if (!f->d_table) TB_HIP(hipMalloc((void **)&f->d_table, sizeof(double) * n));
    f->table_len = n;
    TB_HIP(hipMemcpyAsync(f->d_table, values, sizeof(double) * n, hipMemcpyHostToDevice, f->mesh->dev->stream));
    TB_SYNC_STREAM(f->mesh->dev);
    return TB_OK;
}

// ------------------------------------------------------------------ assembly
// a value array that is rewritten through the library loses its sliced mirror (tb_spmv_mirror): products fall back to the CSR array until the caller binds again
static void mirror_drop(tb_pattern *pat, const double *d_nz)
{
    if (!d_nz) return;
    for (const double *&q : pat->mir_nz) if (q == d_nz) q = nullptr;
}

int tb_assemble_matrix(tb_form *form, tb_pattern *pat, int strategy, double t, double *d_nzval)
{
    TB_REQUIRE(form && pat && (d_nzval || pat->nnz == 0), "tb_assemble_matrix: NULL argument");
    TB_REQUIRE(form->mesh == pat->mesh, "tb_assemble_matrix: form and pattern belong to different meshes");
    TB_REQUIRE(form->kind == TB_FORM_MASS || form->kind == TB_FORM_DIFFUSION, "tb_assemble_matrix: form is not bilinear");
    TB_REQUIRE(form->mesh->ncomp == 1, "tb_assemble_matrix: scalar fields only");
    TB_REQUIRE(strategy >= TB_STRATEGY_ATOMIC && strategy <= TB_STRATEGY_PATCH, "tb_assemble_matrix: unknown strategy %d", strategy);
    TB_HIP(hipSetDevice(form->mesh->dev->id));
    mirror_drop(pat, d_nzval);
    if (form->mesh->n_cells == 0) { // nothing to integrate: the operator is the zero matrix (outputs are overwritten)
        if (pat->nnz) TB_HIP(hipMemsetAsync(d_nzval, 0, sizeof(double) * (size_t)pat->nnz, form->mesh->dev->stream));
        return TB_OK;
    }
    return launch_assemble_matrix(form, pat, strategy, t, d_nzval);
}

int tb_assemble_matrix_pair(tb_form *mass, tb_form *diffusion, tb_pattern *pat, int strategy, double t, double *d_nzval_mass, double *d_nzval_diffusion)
{
    TB_REQUIRE(mass && diffusion && pat && ((d_nzval_mass && d_nzval_diffusion) || pat->nnz == 0), "tb_assemble_matrix_pair: NULL argument");
    TB_REQUIRE(mass->kind == TB_FORM_MASS && diffusion->kind == TB_FORM_DIFFUSION, "tb_assemble_matrix_pair: expects (mass form, diffusion form)");
    TB_REQUIRE(mass->mesh == pat->mesh && diffusion->mesh == pat->mesh, "tb_assemble_matrix_pair: forms and pattern belong to different meshes");
    TB_REQUIRE(d_nzval_mass != d_nzval_diffusion || pat->nnz == 0, "tb_assemble_matrix_pair: the two outputs alias");
    mirror_drop(pat, d_nzval_mass); mirror_drop(pat, d_nzval_diffusion);
    if (strategy == TB_STRATEGY_PATCH && mass->mesh->n_cells > 0 && hex8_patch_applicable(mass, pat) && hex8_patch_applicable(diffusion, pat)) {
        TB_HIP(hipSetDevice(mass->mesh->dev->id));
        const int rc = launch_assemble_hex8_patch(diffusion, mass, pat, t, d_nzval_diffusion, d_nzval_mass);
        if (rc != TB_ERR_UNSUPPORTED) return rc;
    }
    if (strategy == TB_STRATEGY_PATCH && mass->mesh->n_cells > 0 && tet4_patch_applicable(mass, pat) && tet4_patch_applicable(diffusion, pat)) {
        TB_HIP(hipSetDevice(mass->mesh->dev->id));
        const int rc = launch_assemble_tet4_patch(diffusion, mass, pat, t, d_nzval_diffusion, d_nzval_mass);
        if (rc != TB_ERR_UNSUPPORTED) return rc;
    }
    int rc = tb_assemble_matrix(mass, pat, strategy, t, d_nzval_mass);
    if (rc) return rc;
    return tb_assemble_matrix(diffusion, pat, strategy, t, d_nzval_diffusion);
}

int tb_assemble_vector(tb_form *form, int strategy, double t, double *d_b)
{
    TB_REQUIRE(form && d_b, "tb_assemble_vector: NULL argument");
    TB_REQUIRE(form->kind == TB_FORM_SOURCE, "tb_assemble_vector: form is not linear");
    TB_REQUIRE(form->coef.kind != TB_SRC_TABULATED || form->d_table, "tb_assemble_vector: tabulated source without table");
    TB_REQUIRE(strategy >= TB_STRATEGY_ATOMIC && strategy <= TB_STRATEGY_PATCH, "tb_assemble_vector: unknown strategy %d", strategy);
    TB_HIP(hipSetDevice(form->mesh->dev->id));
    if (form->mesh->n_cells == 0) {
        TB_HIP(hipMemsetAsync(d_b, 0, sizeof(double) * (size_t)form->mesh->ndofs, form->mesh->dev->stream));
        return TB_OK;
    }
    return launch_assemble_vector(form, strategy, t, d_b);
}

// ------------------------------------------------------------------ quasi-static hyperelasticity
int tb_hyperelastic_create(tb_mesh *mesh, int qorder, const tb_material *material, tb_form **out)
{
    TB_REQUIRE(mesh && material && out, "tb_hyperelastic_create: NULL argument");
    *out = nullptr;
    TB_REQUIRE(material->kind >= TB_MATERIAL_HOLZAPFEL_OGDEN_2009 && material->kind <= TB_MATERIAL_GUCCIONE_1991, "tb_hyperelastic_create: unknown material kind %d", material->kind);
    TB_REQUIRE(material->reserved >= TB_PENALTY_SIMPLE && material->reserved <= TB_PENALTY_HARTMANN_NEFF_3, "tb_hyperelastic_create: unknown compression penalty %d", material->reserved);
    TB_REQUIRE(mesh->ncomp == 3 && mesh->geom_kind == TB_HEX8, "tb_hyperelastic_create: needs a 3-component field on hexahedra");
    // the kernels address a node's three dofs as consecutive ids (Ferrite: node-major, component-minor, io.jl:233-238)
    for (int64_t i = 0; i < mesh->n_cells * mesh->nb; ++i) {
        const int32_t *d = &mesh->h_cell_dofs[3 * i];
        TB_REQUIRE(d[1] == d[0] + 1 && d[2] == d[0] + 2, "tb_hyperelastic_create: dofs of a node are not consecutive (cell %lld)", (long long)(i / mesh->nb));
    }
    if (qorder == 0) qorder = std::max(2 * kind_order(mesh->field_kind) - 1, 2);
    auto f = std::make_unique<tb_form>();
    f->mesh = mesh; f->kind = TB_FORM_HYPERELASTIC; f->qorder = qorder; f->mat = *material;
    f->mat.fsn_field = nullptr;
    f->act_tension = material->p[9];
    if (material->fsn_field) {
        const int64_t need = mesh->n_cells * 72;
        TB_REQUIRE(material->fsn_field_len == need, "tb_hyperelastic_create: microstructure field needs %lld values, got %lld", (long long)need,
                   (long long)material->fsn_field_len);
        TB_HIP(hipMalloc((void **)&f->d_field, sizeof(double) * need));
        TB_HIP(hipMemcpyAsync(f->d_field, material->fsn_field, sizeof(double) * need, hipMemcpyHostToDevice, mesh->dev->stream));
        TB_SYNC_STREAM(mesh->dev);
    }
    *out = f.release();
    return TB_OK;
}

int tb_hyperelastic_set_active_tension(tb_form *form, double tension, const double *state_field, int64_t len)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC, "tb_hyperelastic_set_active_tension: not a hyperelastic form");
    tb_mesh *m = form->mesh;
    form->act_tension = tension;
    if (!state_field) { hipFree(form->d_act_field); form->d_act_field = nullptr; return TB_OK; }
    TB_REQUIRE(len == m->n_cells * 8, "tb_hyperelastic_set_active_tension: state field needs %lld values (cells × 8), got %lld", (long long)(m->n_cells * 8), (long long)len);
    TB_HIP(hipSetDevice(m->dev->id));
    if (!form->d_act_field && len) TB_HIP(hipMalloc((void **)&form->d_act_field, sizeof(double) * len));
    if (len) {
        TB_HIP(hipMemcpyAsync(form->d_act_field, state_field, sizeof(double) * len, hipMemcpyHostToDevice, m->dev->stream));
        TB_SYNC_STREAM(m->dev);
    }
    return TB_OK;
}

int tb_hyperelastic_set_prestress(tb_form *form, const double *F0inv)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC, "tb_hyperelastic_set_prestress: not a hyperelastic form");
    form->prestressed = F0inv != nullptr;
    if (F0inv) {
        TB_REQUIRE(!form->cond_model, "tb_hyperelastic_set_prestress: not combined with condensed internal variables");
        for (int i = 0; i < 9; ++i) form->prestress_G[i] = F0inv[i];
    }
    return TB_OK;
}

int tb_form_set_cellset(tb_form *form, const int32_t *cells, int64_t n_cells, int index_base)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC, "tb_form_set_cellset: subdomains are implemented for hyperelastic forms");
    TB_REQUIRE(index_base == 0 || index_base == 1, "tb_form_set_cellset: index_base must be 0 or 1");
    TB_REQUIRE(n_cells >= 0 && (cells || n_cells == 0), "tb_form_set_cellset: bad cell list");
    hipFree(form->d_cellset); form->d_cellset = nullptr;
    form->set_colors.reset();
    form->h_cellset.clear();
    if (!cells && n_cells == 0 && index_base == 0 && !form->has_cellset) return TB_OK;
    std::vector<uint8_t> seen(form->mesh->n_cells, 0);
    form->h_cellset.reserve(n_cells);
    for (int64_t i = 0; i < n_cells; ++i) {
        const int64_t c = (int64_t)cells[i] - index_base;
        TB_REQUIRE(c >= 0 && c < form->mesh->n_cells, "tb_form_set_cellset: cell %lld out of range", (long long)cells[i]);
        TB_REQUIRE(!seen[c], "tb_form_set_cellset: cell %lld listed twice", (long long)cells[i]);
        seen[c] = 1;
        form->h_cellset.push_back((int32_t)c);
    }
    form->has_cellset = true;
    form->n_set = n_cells;
    if (n_cells) {
        TB_HIP(hipMalloc((void **)&form->d_cellset, sizeof(int32_t) * n_cells));
        TB_HIP(hipMemcpy(form->d_cellset, form->h_cellset.data(), sizeof(int32_t) * n_cells, hipMemcpyHostToDevice));
    }
    return TB_OK;
}

int tb_form_clear_cellset(tb_form *form)
{
    TB_REQUIRE(form, "tb_form_clear_cellset: NULL form");
    hipFree(form->d_cellset); form->d_cellset = nullptr;
    form->set_colors.reset(); form->h_cellset.clear(); form->has_cellset = false; form->n_set = 0;
    return TB_OK;
}

int tb_form_set_accumulate(tb_form *form, int accumulate)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC, "tb_form_set_accumulate: implemented for hyperelastic forms");
    form->accumulate = accumulate != 0;
    return TB_OK;
}

int tb_hyperelastic_set_condensation(tb_form *form, int sarcomere_model, const double *params, int n_params, double tmax, double local_tol, int local_max_iters)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC, "tb_hyperelastic_set_condensation: not a hyperelastic form");
    if (sarcomere_model < 0) { form->cond_model = 0; return TB_OK; }
    TB_REQUIRE(sarcomere_model == TB_SARCOMERE_RDQ20MF, "tb_hyperelastic_set_condensation: sarcomere model %d has no internal state (TB_SARCOMERE_RDQ20MF)", sarcomere_model);
    TB_REQUIRE(params && n_params == 17, "tb_hyperelastic_set_condensation: RDQ20MF takes 17 parameters");
    TB_REQUIRE(local_tol >= 0.0 && local_max_iters >= 1, "tb_hyperelastic_set_condensation: need tol >= 0 and max_iters >= 1");
    TB_REQUIRE(form->hill == 0, "tb_hyperelastic_set_condensation: Hill frameworks take steady-state sarcomere models only");
    TB_REQUIRE(!form->prestressed, "tb_hyperelastic_set_condensation: not combined with a prestress");
    form->cond_model = sarcomere_model;
    for (int i = 0; i < 17; ++i) form->cond_params[i] = params[i];
    form->cond_tmax = tmax; form->cond_tol = local_tol; form->cond_max_iters = local_max_iters;
    return TB_OK;
}

int tb_hyperelastic_n_quadrature_points(tb_form *form, int64_t *n_points)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC && n_points, "tb_hyperelastic_n_quadrature_points: bad argument");
    const int nq = form->qorder * form->qorder * form->qorder;
    *n_points = form->mesh->n_cells * nq;
    return TB_OK;
}

int tb_hyperelastic_set_internal_state(tb_form *form, double *d_state, const double *d_state_known, double dt)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC, "tb_hyperelastic_set_internal_state: not a hyperelastic form");
    TB_REQUIRE(form->cond_model, "tb_hyperelastic_set_internal_state: the form has no condensed internal variable (tb_hyperelastic_set_condensation)");
    TB_REQUIRE(d_state && d_state_known, "tb_hyperelastic_set_internal_state: NULL state");
    TB_REQUIRE(dt > 0.0, "tb_hyperelastic_set_internal_state: the time step must be positive (got %g)", dt);
    form->d_Q = d_state; form->d_Qknown = d_state_known; form->cond_dt = dt;
    return TB_OK;
}

int tb_hyperelastic_set_previous_solution(tb_form *form, const double *d_u_prev)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC && form->cond_model, "tb_hyperelastic_set_previous_solution: the form has no condensed internal variable");
    form->d_u_prev = d_u_prev;
    if (form->d_qp_buf) { hipFree(form->d_qp_buf); form->d_qp_buf = nullptr; form->d_qp_act = nullptr; } // the per-point layout differs between the two forms
    return TB_OK;
}

int tb_hyperelastic_local_solve_report(tb_form *form, int64_t *n_failed, int32_t *status_host, int64_t len)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC && form->cond_model, "tb_hyperelastic_local_solve_report: the form has no condensed internal variable");
    if (n_failed) *n_failed = form->cond_n_failed;
    if (status_host) {
        int64_t npts = 0;
        tb_hyperelastic_n_quadrature_points(form, &npts);
        TB_REQUIRE(len == npts, "tb_hyperelastic_local_solve_report: status buffer has %lld entries, expected %lld", (long long)len, (long long)npts);
        TB_REQUIRE(form->d_qp_buf, "tb_hyperelastic_local_solve_report: no assembly has run yet");
        TB_HIP(hipMemcpy(status_host, (const int32_t *)(form->d_qp_buf + (form->d_u_prev ? 14 : 4) * npts), sizeof(int32_t) * npts, hipMemcpyDeviceToHost));
    }
    return TB_OK;
}

static int validate_hill(const tb_hill *h, const char *who)
{
    TB_REQUIRE(h->framework >= TB_HILL_NONE && h->framework <= TB_HILL_EXTENDED, "%s: unknown framework %d", who, h->framework);
    if (h->framework == TB_HILL_NONE) return TB_OK;
    TB_REQUIRE(h->active_energy == TB_ACTIVE_SIMPLE_SPRING || (h->active_energy >= TB_MATERIAL_HOLZAPFEL_OGDEN_2009 && h->active_energy <= TB_MATERIAL_GUCCIONE_1991),
               "%s: unknown active spring %d", who, h->active_energy);
    TB_REQUIRE(h->active_penalty >= TB_PENALTY_SIMPLE && h->active_penalty <= TB_PENALTY_HARTMANN_NEFF_3, "%s: unknown penalty %d", who, h->active_penalty);
    TB_REQUIRE(h->adg_kind >= TB_ADG_GMK && h->adg_kind <= TB_ADG_RLRSQ, "%s: unknown active deformation gradient model %d", who, h->adg_kind);
    TB_REQUIRE(h->sarcomere_kind == TB_SARCOMERE_PELCE_SUN_LANGEVELD_1995 || h->sarcomere_kind == TB_SARCOMERE_CONSTANT_STRETCH,
               "%s: unknown sarcomere model %d", who, h->sarcomere_kind);
    return TB_OK;
}
static void store_hill(tb_form *form, const tb_hill *h)
{
    form->hill = h ? h->framework : 0;
    if (!form->hill) return;
    form->act_energy = h->active_energy; form->act_penalty = h->active_penalty; form->adg = h->adg_kind; form->sarc = h->sarcomere_kind;
    for (int i = 0; i < 12; ++i) form->act_p[i] = h->active_p[i];
    form->hill_kappa = h->sheetlet_part; form->sarc_p[0] = h->sarcomere_p[0]; form->sarc_p[1] = h->sarcomere_p[1];
}

int tb_hyperelastic_set_hill(tb_form *form, const tb_hill *hill)
{
    TB_REQUIRE(form && form->kind == TB_FORM_HYPERELASTIC, "tb_hyperelastic_set_hill: not a hyperelastic form");
    if (hill) { int rc = validate_hill(hill, "tb_hyperelastic_set_hill"); if (rc) return rc; }
    store_hill(form, hill);
    return TB_OK;
}

int tb_host_material_eval_hill(const tb_material *material, const tb_hill *hill, double activation, const double *F, double *psi, double *P, double *A)
{
    TB_REQUIRE(material && F, "tb_host_material_eval_hill: NULL argument");
    TB_REQUIRE(material->kind >= TB_MATERIAL_HOLZAPFEL_OGDEN_2009 && material->kind <= TB_MATERIAL_GUCCIONE_1991, "tb_host_material_eval_hill: unknown material kind %d", material->kind);
    TB_REQUIRE(material->reserved >= TB_PENALTY_SIMPLE && material->reserved <= TB_PENALTY_HARTMANN_NEFF_3, "tb_host_material_eval_hill: unknown compression penalty %d", material->reserved);
    if (hill) { int rc = validate_hill(hill, "tb_host_material_eval_hill"); if (rc) return rc; }
    tb_form tmp;
    tmp.mat = *material;
    tmp.mat.p[9] = activation;
    store_hill(&tmp, hill);
    return host_material_eval_form(&tmp, F, psi, P, A);
}

int tb_residual(tb_form *form, int strategy, const double *d_u, double t, double *d_r)
{
    (void)t;
    TB_REQUIRE(form && d_u && d_r, "tb_residual: NULL argument");
    TB_REQUIRE(form->kind == TB_FORM_HYPERELASTIC, "tb_residual: form is not nonlinear");
    TB_HIP(hipSetDevice(form->mesh->dev->id));
    if (form->mesh->n_cells == 0) {
        TB_HIP(hipMemsetAsync(d_r, 0, sizeof(double) * (size_t)form->mesh->ndofs, form->mesh->dev->stream));
        return TB_OK;
    }
    return launch_hyperelastic(form, nullptr, strategy, d_u, nullptr, d_r);
}

int tb_linearize(tb_form *form, tb_pattern *pat, int strategy, const double *d_u, double t, double *d_nzval, double *d_r)
{
    (void)t;
    TB_REQUIRE(form && pat && d_u && (d_nzval || pat->nnz == 0), "tb_linearize: NULL argument");
    TB_REQUIRE(form->kind == TB_FORM_HYPERELASTIC, "tb_linearize: form is not nonlinear");
    TB_REQUIRE(form->mesh == pat->mesh, "tb_linearize: form and pattern belong to different meshes");
    TB_HIP(hipSetDevice(form->mesh->dev->id));
    if (form->mesh->n_cells == 0) {
        if (pat->nnz) TB_HIP(hipMemsetAsync(d_nzval, 0, sizeof(double) * (size_t)pat->nnz, form->mesh->dev->stream));
        if (d_r) TB_HIP(hipMemsetAsync(d_r, 0, sizeof(double) * (size_t)form->mesh->ndofs, form->mesh->dev->stream));
        return TB_OK;
    }
    return launch_hyperelastic(form, pat, strategy, d_u, d_nzval, d_r);
}

int tb_facet_form_create(tb_mesh *mesh, int bc_kind, double param, int facet_qpoints, const int32_t *facets, int64_t n_facets,
                         int index_base, tb_form **out)
{
    TB_REQUIRE(mesh && out && (facets || n_facets == 0), "tb_facet_form_create: NULL argument");
    *out = nullptr;
    TB_REQUIRE(bc_kind >= TB_BC_ROBIN && bc_kind <= TB_BC_PRESSURE_FIELD, "tb_facet_form_create: unknown boundary condition %d", bc_kind);
    TB_REQUIRE(mesh->ncomp == 3 && mesh->geom_kind == TB_HEX8, "tb_facet_form_create: needs a 3-component field on hexahedra");
    TB_REQUIRE(index_base == 0 || index_base == 1, "tb_facet_form_create: index_base must be 0 or 1");
    TB_REQUIRE(n_facets >= 0, "tb_facet_form_create: negative facet count");
    if (facet_qpoints == 0) facet_qpoints = kind_order(mesh->field_kind);
    TB_REQUIRE(facet_qpoints >= 1 && facet_qpoints <= 3, "tb_facet_form_create: 1…3 Gauss points per facet direction (got %d)", facet_qpoints);
    for (int64_t i = 0; i < mesh->n_cells * mesh->nb; ++i) {
        const int32_t *d = &mesh->h_cell_dofs[3 * i];
        TB_REQUIRE(d[1] == d[0] + 1 && d[2] == d[0] + 2, "tb_facet_form_create: dofs of a node are not consecutive (cell %lld)", (long long)(i / mesh->nb));
    }
    std::vector<int32_t> fl((size_t)2 * n_facets);
    for (int64_t i = 0; i < n_facets; ++i) {
        const int32_t c = facets[2 * i] - index_base, lf = facets[2 * i + 1] - index_base;
        TB_REQUIRE(c >= 0 && c < mesh->n_cells && lf >= 0 && lf < 6, "tb_facet_form_create: facet %lld = (%d, %d) out of range", (long long)i, facets[2 * i], facets[2 * i + 1]);
        fl[2 * i] = c; fl[2 * i + 1] = lf;
    }
    auto f = std::make_unique<tb_form>();
    f->mesh = mesh; f->kind = TB_FORM_FACET; f->bc_kind = bc_kind; f->bc_param = param; f->facet_q = facet_qpoints; f->n_facets = n_facets;
    TB_HIP(hipSetDevice(mesh->dev->id));
    int rc = upload(mesh->dev, fl, &f->d_facets);
    if (rc) return rc;
    *out = f.release();
    return TB_OK;
}

int tb_facet_form_set_field(tb_form *form, const double *field, int64_t len)
{
    TB_REQUIRE(form && form->kind == TB_FORM_FACET && form->bc_kind == TB_BC_PRESSURE_FIELD, "tb_facet_form_set_field: not a PressureFieldBC form");
    tb_mesh *m = form->mesh;
    if (!field) { hipFree(form->d_field); form->d_field = nullptr; return TB_OK; }
    TB_REQUIRE(len == m->n_cells * 8, "tb_facet_form_set_field: needs %lld values (cells × 8), got %lld", (long long)(m->n_cells * 8), (long long)len);
    TB_HIP(hipSetDevice(m->dev->id));
    if (!form->d_field) TB_HIP(hipMalloc((void **)&form->d_field, sizeof(double) * len));
    TB_HIP(hipMemcpyAsync(form->d_field, field, sizeof(double) * len, hipMemcpyHostToDevice, m->dev->stream));
    TB_SYNC_STREAM(m->dev);
    return TB_OK;
}

int tb_facet_form_set_param(tb_form *form, double param)
{
    TB_REQUIRE(form && form->kind == TB_FORM_FACET, "tb_facet_form_set_param: not a facet form");
    form->bc_param = param;
    return TB_OK;
}

int tb_facet_assemble(tb_form *form, tb_pattern *pat, const double *d_u, double t, double *d_nzval, double *d_r)
{
    (void)t;
    TB_REQUIRE(form && d_u && (d_nzval || d_r), "tb_facet_assemble: NULL argument");
    TB_REQUIRE(form->kind == TB_FORM_FACET, "tb_facet_assemble: form is not a weak boundary condition");
    TB_REQUIRE(!d_nzval || (pat && pat->mesh == form->mesh), "tb_facet_assemble: the tangent needs the pattern of the form's mesh");
    if (form->n_facets == 0) return TB_OK;
    TB_HIP(hipSetDevice(form->mesh->dev->id));
    return launch_facets(form, pat, d_u, d_nzval, d_r);
}

int tb_host_material_eval(const tb_material *material, const double *F, double *psi, double *P, double *A)
{
    TB_REQUIRE(material && F, "tb_host_material_eval: NULL argument");
    TB_REQUIRE(material->kind >= TB_MATERIAL_HOLZAPFEL_OGDEN_2009 && material->kind <= TB_MATERIAL_GUCCIONE_1991, "tb_host_material_eval: unknown material kind %d", material->kind);
    TB_REQUIRE(material->reserved >= TB_PENALTY_SIMPLE && material->reserved <= TB_PENALTY_HARTMANN_NEFF_3, "tb_host_material_eval: unknown compression penalty %d", material->reserved);
    return host_material_eval(material, F, psi, P, A);
}

// ------------------------------------------------------------------ sarcomere
int tb_sarcomere_model_info(int model, int *n_states, int *n_params)
{
    TB_REQUIRE(model == TB_SARCOMERE_RDQ20MF, "tb_sarcomere_model_info: model %d has no internal state (only TB_SARCOMERE_RDQ20MF)", model);
    if (n_states) *n_states = 20;
    if (n_params) *n_params = 17;
    return TB_OK;
}

int tb_sarcomere_step(tb_device *dev, int model, const double *params, int n_params, double *d_state, int64_t n_points,
                      const double *d_stretch, const double *d_velocity, const double *d_calcium, double stretch, double velocity,
                      double calcium, double t, double dt, int substeps, int rate_independent, double *d_tension, double *d_stiffness)
{
    (void)t; // the model is autonomous given its inputs (the reference passes t through and never uses it)
    TB_REQUIRE(dev && params, "tb_sarcomere_step: NULL argument");
    TB_REQUIRE(model == TB_SARCOMERE_RDQ20MF, "tb_sarcomere_step: unknown sarcomere model %d", model);
    TB_REQUIRE(n_params == 17, "tb_sarcomere_step: RDQ20MF takes 17 parameters (got %d)", n_params);
    TB_REQUIRE(n_points >= 0 && substeps >= 1, "tb_sarcomere_step: n_points must be >= 0 and substeps >= 1");
    if (n_points == 0) return TB_OK;
    TB_REQUIRE(d_state, "tb_sarcomere_step: NULL state");
    TB_HIP(hipSetDevice(dev->id));
    return launch_sarcomere(dev, params, d_state, n_points, d_stretch, d_velocity, d_calcium, stretch, velocity, calcium, dt, substeps,
                            rate_independent, d_tension, d_stiffness);
}

int tb_host_sarcomere_eval(int model, const double *params, int n_params, const double *state, double stretch, double velocity,
                           double calcium, double *dstate, double *tension, double *stiffness)
{
    TB_REQUIRE(params && state, "tb_host_sarcomere_eval: NULL argument");
    TB_REQUIRE(model == TB_SARCOMERE_RDQ20MF && n_params == 17, "tb_host_sarcomere_eval: RDQ20MF with 17 parameters expected");
    host_sarcomere_eval(params, state, stretch, velocity, calcium, dstate, tension, stiffness);
    return TB_OK;
}

int tb_host_sarcomere_derivatives(int model, const double *params, int n_params, const double *state, double stretch, double velocity, double calcium,
                                  int analytic, double *drhs_dstate, double *drhs_dstretch, double *drhs_dvelocity, double *rhs)
{
    TB_REQUIRE(params && state && drhs_dstate && drhs_dstretch && drhs_dvelocity && rhs, "tb_host_sarcomere_derivatives: NULL argument");
    TB_REQUIRE(model == TB_SARCOMERE_RDQ20MF && n_params == 17, "tb_host_sarcomere_derivatives: RDQ20MF with 17 parameters expected");
    host_sarcomere_derivatives(params, state, stretch, velocity, calcium, analytic, drhs_dstate, drhs_dstretch, drhs_dvelocity, rhs);
    return TB_OK;
}

int tb_sarcomere_implicit_step(tb_device *dev, int model, const double *params, int n_params, double *d_state, const double *d_state_known,
                               int64_t n_points, const double *d_stretch, const double *d_velocity, const double *d_calcium, double stretch,
                               double velocity, double calcium, double dt, double tol, int max_iters, double *d_dstate_dstretch,
                               double *d_dstate_dvelocity, int32_t *d_status, int64_t *n_failed)
{
    TB_REQUIRE(dev && params, "tb_sarcomere_implicit_step: NULL argument");
    TB_REQUIRE(model == TB_SARCOMERE_RDQ20MF && n_params == 17, "tb_sarcomere_implicit_step: RDQ20MF with 17 parameters expected");
    TB_REQUIRE(n_points >= 0 && dt > 0.0 && tol >= 0.0 && max_iters >= 1, "tb_sarcomere_implicit_step: need n_points >= 0, dt > 0, tol >= 0, max_iters >= 1");
    TB_REQUIRE(!d_dstate_dvelocity || d_dstate_dstretch, "tb_sarcomere_implicit_step: the velocity corrector is computed together with the stretch corrector");
    if (n_failed) *n_failed = 0;
    if (n_points == 0) return TB_OK;
    TB_REQUIRE(d_state && d_state_known, "tb_sarcomere_implicit_step: NULL state");
    TB_HIP(hipSetDevice(dev->id));
    return launch_sarcomere_implicit(dev, params, d_state, d_state_known, n_points, d_stretch, d_velocity, d_calcium, stretch, velocity, calcium, dt, tol,
                                     max_iters, 0.0, d_dstate_dstretch, d_dstate_dvelocity, nullptr, 2, d_status, n_failed, d_dstate_dstretch != nullptr);
}

int tb_host_sarcomere_local_solve(int model, const double *params, int n_params, double *state, const double *state_known, double stretch,
                                  double velocity, double calcium, double dt, double tol, int max_iters, double *dstate_dstretch,
                                  double *dstate_dvelocity, int *status, int *iters, double *resnorm)
{
    TB_REQUIRE(params && state && state_known, "tb_host_sarcomere_local_solve: NULL argument");
    TB_REQUIRE(model == TB_SARCOMERE_RDQ20MF && n_params == 17, "tb_host_sarcomere_local_solve: RDQ20MF with 17 parameters expected");
    TB_REQUIRE(dt > 0.0 && max_iters >= 1, "tb_host_sarcomere_local_solve: need dt > 0 and max_iters >= 1");
    TB_REQUIRE(!dstate_dvelocity || dstate_dstretch, "tb_host_sarcomere_local_solve: the velocity corrector is computed together with the stretch corrector");
    const int code = host_sarcomere_local_solve(params, state, state_known, stretch, velocity, calcium, dt, tol, max_iters, dstate_dstretch, dstate_dvelocity,
                                                iters, resnorm);
    if (status) *status = code;
    return TB_OK;
}

// ------------------------------------------------------------------ reaction
int tb_cell_model_info(int model, int *n_states, int *n_params, int *phi_index)
{
    int ns, np, pi = 0;
    switch (model) {
    case TB_CELL_FHN: ns = 2; np = 6; break;
    case TB_CELL_ALIEV_PANFILOV: ns = 2; np = 6; pi = 1; break; // state order (s, φₘ): aliev-panfilov.jl:13
    case TB_CELL_PCG2019: ns = 7; np = 36; break;
    case TB_CELL_TT06: ns = 19; np = 48; break;
    case TB_CELL_FHN_HETEROGENEOUS: ns = 2; np = 8; break;
    case TB_CELL_ORD11: ns = 41; np = 17; break;
    default: set_error("unknown cell model %d", model); return TB_ERR_BAD_ARG;
    }
    if (n_states) *n_states = ns;
    if (n_params) *n_params = np;
    if (phi_index) *phi_index = pi;
    return TB_OK;
}

int tb_cell_model_defaults(int model, double *params, double *u0)
{
    TB_REQUIRE(params, "tb_cell_model_defaults: params is NULL");
    switch (model) {
    case TB_CELL_FHN: { // src/modeling/cells/fhn.jl:6-13
        const double d[6] = {0.1, 0.5, 1.0, 0.0, 0.01, 1.0};
        memcpy(params, d, sizeof d);
        if (u0) u0[0] = u0[1] = 0.0;
        return TB_OK;
    }
    case TB_CELL_FHN_HETEROGENEOUS: { // docs/src/literate-howto/custom-ep-cell-model.jl:8-15: a, b, c, d, e ≡ 0.01 (no gradient)
        const double d[8] = {0.1, 0.5, 1.0, 0.0, 0.01, 0.0, 0.0, 0.0};
        memcpy(params, d, sizeof d);
        if (u0) u0[0] = u0[1] = 0.0;
        return TB_OK;
    }
    case TB_CELL_ALIEV_PANFILOV: { // src/modeling/cells/aliev-panfilov.jl:1-8
        const double d[6] = {1.0 / 12.9, 8.0, 0.05, 0.002, 0.2, 0.3};
        memcpy(params, d, sizeof d);
        if (u0) u0[0] = u0[1] = 0.0;
        return TB_OK;
    }
    case TB_CELL_PCG2019: { // src/modeling/cells/pcg2019.jl:4-48 (field order), :137-152 (initial state)
        const double d[36] = {12.0, -52.244, 6.5472, 0.12, -78.7, 5.93, 0.799163, 6.80738, 0.73893, -91.9655, 12.4997, 0.1688,
                              14.3116, 11.462, -47.9286, 4.9314, 9.90669, 0.11503, 0.7, 4.3, -15.7, 4.6, 30.0, 0.056,
                              -26.6, 6.5, 334.0, -49.6, 23.5, 0.008, 24.6, 12.1, 628.0, 65.0, -85.0, 50.0};
        memcpy(params, d, sizeof d);
        if (u0) {
            auto sig = [](double phi, double E, double k, double sign) { return 1.0 / (1.0 + std::exp(sign * (phi - E) / k)); };
            u0[0] = d[34];
            u0[1] = sig(u0[0], d[4], d[5], 1.0);    // h
            u0[2] = sig(u0[0], d[1], d[2], -1.0);   // m
            u0[3] = sig(u0[0], d[20], d[21], 1.0);  // f
            u0[4] = sig(u0[0], d[14], d[15], 1.0);  // s
            u0[5] = sig(u0[0], d[30], d[31], -1.0); // xs
            u0[6] = sig(u0[0], d[24], d[25], -1.0); // xr
        }
        return TB_OK;
    }
    case TB_CELL_TT06: { // ten Tusscher & Panfilov 2006, epicardial set; resting state of the authors' implementation
        const double d[48] = {14.838, 5.405, 0.153, 0.392, 0.294, 3.98e-5, 0.00029, 0.000592, 0.1238, 0.0146, 2.724, 1000.0,
                              5.4, 2.0, 140.0, 0.016404, 0.001094, 0.00005468, 0.2, 0.001, 10.0, 0.3, 0.4, 0.00025,
                              0.006375, 0.00025, 0.102, 0.15, 0.045, 0.060, 0.005, 1.5, 2.5, 1.0, 0.00036, 0.0038,
                              0.185, 0.03, 1.0, 40.0, 87.5, 1.38, 0.1, 0.35, 0.0005, 8314.472, 310.0, 96485.3415};
        memcpy(params, d, sizeof d);
        if (u0) {
            const double s[19] = {-86.2, 0.00007, 1.3, 0.00007, 7.67, 138.3, 0.0, 0.75, 0.75, 0.0, 1.0, 0.0, 0.0, 1.0, 0.0, 1.0, 1.0, 1.0, 1.0};
            memcpy(u0, s, sizeof s);
        }
        return TB_OK;
    }
    case TB_CELL_ORD11: { // O'Hara–Rudy 2011: unit conductance scalings, the paper's extracellular concentrations, endocardial cell; initial state of its supplement
        const double d[17] = {1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, 140.0, 1.8, 5.4, 0.0};
        memcpy(params, d, sizeof d);
        if (u0) {
            const double s[41] = {-87.0, 7.0, 7.0, 145.0, 145.0, 1.0e-4, 1.0e-4, 1.2, 1.2, 0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 1.0, 0.0, 1.0, 1.0, 0.0, 1.0, 1.0,
                                  0.0, 1.0, 1.0, 1.0, 1.0, 1.0, 0.0, 1.0, 1.0, 0.0, 0.0, 0.0, 0.0, 1.0, 0.0, 0.0, 0.0};
            memcpy(u0, s, sizeof s);
        }
        return TB_OK;
    }
    }
    set_error("unknown cell model %d", model);
    return TB_ERR_BAD_ARG;
}

static int reaction_step(const char *who, tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du,
                         int64_t n_points, int n_states, int layout, double t, double dt, int substeps, double threshold, double *rmax,
                         const float *d_x = nullptr, int sdim = 0)
{
    TB_REQUIRE(model != TB_CELL_FHN_HETEROGENEOUS || d_x || n_points == 0, "%s: this cell model reads the point coordinate — pass d_x (tb_reaction_step_x)", who);
    TB_REQUIRE(!d_x || (sdim >= 1 && sdim <= 3), "%s: sdim must be 1, 2 or 3 (got %d)", who, sdim);
    TB_REQUIRE(dev && params && (d_u || n_points == 0), "%s: NULL argument", who);
    int ns, np;
    int rc = tb_cell_model_info(model, &ns, &np, nullptr);
    if (rc) return rc;
    TB_REQUIRE(n_states == ns, "%s: model has %d states, caller says %d", who, ns, n_states);
    TB_REQUIRE(n_params == np, "%s: model has %d parameters, caller passed %d", who, np, n_params);
    TB_REQUIRE(layout == TB_LAYOUT_SOA || layout == TB_LAYOUT_AOS, "%s: unknown layout %d", who, layout);
    TB_REQUIRE(n_points >= 0, "%s: negative point count", who);
    if (n_points == 0) { if (rmax) *rmax = -__builtin_huge_val(); return TB_OK; }
    TB_HIP(hipSetDevice(dev->id));
    return launch_reaction(dev, model, params, n_params, d_u, d_du, n_points, layout, t, dt, substeps, threshold, rmax, d_x, sdim);
}

int tb_reaction_step_x(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du, int64_t n_points, int n_states,
                       int layout, const float *d_x, int sdim, double t, double dt, int substeps, double threshold)
{
    return reaction_step("tb_reaction_step_x", dev, model, params, n_params, d_u, d_du, n_points, n_states, layout, t, dt, substeps, threshold, nullptr, d_x, sdim);
}

int tb_reaction_step(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du,
                     int64_t n_points, int n_states, int layout, double t, double dt, int substeps, double threshold)
{
    return reaction_step("tb_reaction_step", dev, model, params, n_params, d_u, d_du, n_points, n_states, layout, t, dt, substeps, threshold, nullptr);
}

int tb_reaction_step_rtc(tb_device *dev, int model, const double *params, int n_params, double *d_u, double *d_du,
                         int64_t n_points, int n_states, int layout, double t, double dt, int substeps, double threshold,
                         double *rmax)
{
    TB_REQUIRE(rmax, "tb_reaction_step_rtc: NULL rmax");
    return reaction_step("tb_reaction_step_rtc", dev, model, params, n_params, d_u, d_du, n_points, n_states, layout, t, dt, substeps, threshold, rmax);
}

int tb_reaction_step_rl(tb_device *dev, int model, const double *params, int n_params, double *d_u, int64_t n_points, int n_states,
                        int layout, double t, double dt)
{
    TB_REQUIRE(dev && params && (d_u || n_points == 0), "tb_reaction_step_rl: NULL argument");
    int ns, np;
    int rc = tb_cell_model_info(model, &ns, &np, nullptr);
    if (rc) return rc;
    TB_REQUIRE(n_states == ns && n_params == np, "tb_reaction_step_rl: model has %d states / %d parameters, caller says %d / %d", ns, np, n_states, n_params);
    TB_REQUIRE(layout == TB_LAYOUT_SOA || layout == TB_LAYOUT_AOS, "tb_reaction_step_rl: unknown layout %d", layout);
    TB_REQUIRE(n_points >= 0, "tb_reaction_step_rl: negative point count");
    if (n_points == 0) return launch_reaction_rl(dev, model, params, n_params, d_u, 0, layout, t, dt);
    TB_HIP(hipSetDevice(dev->id));
    return launch_reaction_rl(dev, model, params, n_params, d_u, n_points, layout, t, dt);
}

// ------------------------------------------------------------------ algebra
int tb_heat_matrix(tb_device *dev, int64_t nnz, const double *d_Mnz, const double *d_Knz, double dt, double *d_Anz)
{
    TB_REQUIRE(dev && d_Mnz && d_Knz && d_Anz && nnz >= 0, "tb_heat_matrix: bad argument");
    if (!nnz) return TB_OK;
    mirror_drop_any(dev, d_Anz, (size_t)nnz * sizeof(double));
    return launch_heat_matrix(dev, nnz, d_Mnz, d_Knz, dt, d_Anz);
}

int tb_spmv_csr(tb_pattern *pat, const double *d_nzval, const double *d_x, double alpha, double beta, double *d_y)
{
    TB_REQUIRE(pat && d_nzval && d_x && d_y, "tb_spmv_csr: NULL argument");
    TB_REQUIRE(d_x != d_y, "tb_spmv_csr: x and y alias");
    return launch_spmv(pat, d_nzval, d_x, alpha, beta, d_y);
}

int tb_cg_solve(tb_pattern *pat, const double *d_Anz, const double *d_b, double *d_x, double rtol, double atol, int maxiter, int jacobi,
                int *iters, double *resnorm)
{
    TB_REQUIRE(pat && d_Anz && d_b && d_x, "tb_cg_solve: NULL argument");
    TB_REQUIRE(rtol >= 0 && atol >= 0 && maxiter >= 0, "tb_cg_solve: negative tolerance or iteration limit");
    TB_REQUIRE(jacobi >= 0 && jacobi <= TB_JACOBI_REUSE, "tb_cg_solve: jacobi must be 0, 1 or TB_JACOBI_REUSE");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_cg(pat, d_Anz, d_b, d_x, rtol, atol, maxiter, jacobi, iters, resnorm);
}

int tb_cg_solve_from_residual(tb_pattern *pat, const double *d_Anz, const double *d_r0, double *d_x, double rtol, double atol, int maxiter, int jacobi,
                              int *iters, double *resnorm)
{
    TB_REQUIRE(pat && d_Anz && d_r0 && d_x, "tb_cg_solve_from_residual: NULL argument");
    TB_REQUIRE(rtol >= 0 && atol >= 0 && maxiter >= 0, "tb_cg_solve_from_residual: negative tolerance or iteration limit");
    TB_REQUIRE(jacobi >= 0 && jacobi <= TB_JACOBI_REUSE, "tb_cg_solve_from_residual: jacobi must be 0, 1 or TB_JACOBI_REUSE");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_cg(pat, d_Anz, d_r0, d_x, rtol, atol, maxiter, jacobi, iters, resnorm, true);
}

int tb_pcg_solve(tb_pattern *pat, const double *d_Anz, const double *d_b, double *d_x, double rtol, double atol, int maxiter, int precond, int partsize,
                 int *iters, double *resnorm)
{
    TB_REQUIRE(pat && d_Anz && d_b && d_x, "tb_pcg_solve: NULL argument");
    TB_REQUIRE(rtol >= 0 && atol >= 0 && maxiter >= 0, "tb_pcg_solve: negative tolerance or iteration limit");
    TB_REQUIRE(precond >= TB_PRECOND_NONE && precond <= TB_PRECOND_CHEBYSHEV, "tb_pcg_solve: unknown preconditioner %d", precond);
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    if (precond == TB_PRECOND_CHEBYSHEV) {
        TB_REQUIRE(partsize >= 1 && partsize <= 64, "tb_pcg_solve: Chebyshev degree must be in 1..64 (got %d)", partsize);
        return launch_pcg_chebyshev(pat, d_Anz, d_b, d_x, rtol, atol, maxiter, partsize, iters, resnorm);
    }
    if (precond != TB_PRECOND_L1GS) return launch_cg(pat, d_Anz, d_b, d_x, rtol, atol, maxiter, precond == TB_PRECOND_JACOBI, iters, resnorm);
    TB_REQUIRE(partsize >= 1 && partsize <= 1024, "tb_pcg_solve: partition size must be in 1..1024 (got %d)", partsize);
    return launch_pcg_l1gs(pat, d_Anz, d_b, d_x, rtol, atol, maxiter, partsize, iters, resnorm);
}

int tb_l1gs_apply(tb_pattern *pat, const double *d_Anz, int partsize, int sweep, const double *d_r, double *d_z)
{
    TB_REQUIRE(pat && d_Anz && d_r && d_z, "tb_l1gs_apply: NULL argument");
    TB_REQUIRE(partsize >= 1 && partsize <= 1024, "tb_l1gs_apply: partition size must be in 1..1024 (got %d)", partsize);
    TB_REQUIRE(sweep == TB_SWEEP_FORWARD || sweep == TB_SWEEP_SYMMETRIC, "tb_l1gs_apply: sweep must be TB_SWEEP_FORWARD or TB_SWEEP_SYMMETRIC");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    if (!pat->d_pcg_ws) TB_HIP(hipMalloc((void **)&pat->d_pcg_ws, sizeof(double) * (5 * pat->n_rows + 8)));
    double *dtl = pat->d_pcg_ws + 4 * pat->n_rows;
    int rc = launch_l1gs_setup(pat, d_Anz, partsize, dtl);
    if (rc) return rc;
    return launch_l1gs_apply(pat, d_Anz, dtl, partsize, sweep == TB_SWEEP_SYMMETRIC, d_r, d_z);
}

int tb_gmres_solve(tb_pattern *pat, const double *d_Anz, const double *d_b, double *d_x, double rtol, double atol, int maxiter, int restart,
                   int jacobi, int *iters, double *resnorm)
{
    TB_REQUIRE(pat && d_Anz && d_b && d_x, "tb_gmres_solve: NULL argument");
    TB_REQUIRE(rtol >= 0 && atol >= 0 && maxiter >= 0, "tb_gmres_solve: negative tolerance or iteration limit");
    TB_REQUIRE(restart >= 1 && restart <= 1000, "tb_gmres_solve: restart must be in 1..1000 (got %d)", restart);
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_gmres(pat, d_Anz, d_b, d_x, rtol, atol, maxiter, restart, jacobi, iters, resnorm);
}

int tb_solver_last_tolerance(const tb_pattern *pat, double *tol)
{
    TB_REQUIRE(pat && tol, "tb_solver_last_tolerance: NULL argument");
    *tol = pat->last_tol;
    return TB_OK;
}

int tb_axpy(tb_device *dev, int64_t n, double a, const double *d_x, double *d_y)
{
    TB_REQUIRE(dev && d_x && d_y && n >= 0, "tb_axpy: bad argument");
    if (!n) return TB_OK;
    mirror_drop_any(dev, d_y, (size_t)n * sizeof(double));
    return launch_axpy(dev, n, a, d_x, d_y);
}

int tb_dot(tb_device *dev, int64_t n, const double *d_x, const double *d_y, double *result)
{
    TB_REQUIRE(dev && result && n >= 0 && ((d_x && d_y) || n == 0), "tb_dot: bad argument");
    return launch_dot(dev, n, d_x, d_y, result);
}

int tb_cgd_dot(tb_device *dev, int64_t n, const double *d_w, const double *d_a, const double *d_b, double *d_out)
{
    TB_REQUIRE(dev && d_out && n >= 0 && ((d_a && d_b) || n == 0), "tb_cgd_dot: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return launch_cgd_dot(dev, n, d_w, d_a, d_b, d_out);
}

int tb_cgd_update(tb_device *dev, int64_t n, const double *d_w, const double *d_dinv, const double *d_p, const double *d_Ap, double *d_x, double *d_r,
                  const double *d_rz, const double *d_pAp, double *d_out3)
{
    TB_REQUIRE(dev && d_rz && d_pAp && d_out3 && n >= 0 && ((d_p && d_Ap && d_x && d_r) || n == 0), "tb_cgd_update: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return launch_cgd_update(dev, n, d_w, d_dinv, d_p, d_Ap, d_x, d_r, d_rz, d_pAp, d_out3);
}

int tb_cgd_direction(tb_device *dev, int64_t n, const double *d_dinv, const double *d_r, double *d_p, const double *d_rz, const double *d_rz_new)
{
    TB_REQUIRE(dev && d_rz && d_rz_new && n >= 0 && ((d_r && d_p) || n == 0), "tb_cgd_direction: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return launch_cgd_direction(dev, n, d_dinv, d_r, d_p, d_rz, d_rz_new);
}

int tb_cgd_rotate(tb_device *dev, double *d_S)
{
    TB_REQUIRE(dev && d_S, "tb_cgd_rotate: NULL argument");
    TB_HIP(hipSetDevice(dev->id));
    return launch_cgd_rotate(dev, d_S);
}

// One whole iteration of the device CG of a sub-domain WITHOUT shared dofs (one rank, or an isolated part): the four launches of
// tb_spmv_csr_dot → tb_cgd_update → tb_cgd_direction → tb_cgd_rotate issued from one call.  Same kernels, same results; what it saves is the host's
// per-call cost, which on a thin slab (0.095 ms of kernels per iteration at 216×216×27) is a third of the iteration when the host is an interpreter.
int tb_cgd_iteration(tb_pattern *pat, const double *d_nzval, const double *d_dinv, double *d_x, double *d_r, double *d_p, double *d_Ap, double *d_S)
{
    TB_REQUIRE(pat && d_nzval && d_dinv && d_x && d_r && d_p && d_Ap && d_S, "tb_cgd_iteration: NULL argument");
    tb_device *dev = pat->mesh->dev;
    TB_HIP(hipSetDevice(dev->id));
    const int64_t n = pat->n_rows;
    (void)n;
    return launch_cgd_iteration(pat, d_nzval, d_dinv, d_x, d_r, d_p, d_Ap, d_S);
}

int tb_extract_diagonal(tb_pattern *pat, const double *d_nzval, double *d_diag)
{
    TB_REQUIRE(pat && ((d_nzval && d_diag) || pat->n_rows == 0), "tb_extract_diagonal: NULL argument");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_extract_diagonal(pat, d_nzval, d_diag);
}

int tb_pattern_spmv_plan(tb_pattern *pat, int64_t *out2)
{
    TB_REQUIRE(pat && out2, "tb_pattern_spmv_plan: NULL argument");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    int rc = spmv_plans(pat);
    if (rc) return rc;
    out2[0] = pat->n_sig; out2[1] = pat->sig_entries;
    return TB_OK;
}

int tb_pattern_patch_stats(tb_pattern *pat, int64_t *out6)
{
    TB_REQUIRE(pat && out6, "tb_pattern_patch_stats: NULL argument");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    int rc = ensure_patch_plans(pat->mesh, nullptr); // the mesh-level plan as the kernels last (re)fitted it; not the per-pattern extensions
    if (rc) return rc;
    const PatchPlan *pp = pat->mesh->patches.get();
    out6[0] = pp->n_patches; out6[1] = pp->total_elems; out6[2] = pat->mesh->n_cells; out6[3] = pp->max_elems; out6[4] = pp->max_rows;
    out6[5] = pat->patch_fused ? (int64_t)pat->patch_fused->max_lds_entries * 8 : (pat->patch_mat ? (int64_t)pat->patch_mat->max_lds_entries * 8 : 0);
    return TB_OK;
}

int tb_gather_indexed(tb_device *dev, int64_t n, const double *d_vec, const int32_t *d_idx, double *d_out)
{
    TB_REQUIRE(dev && n >= 0 && ((d_vec && d_idx && d_out) || n == 0), "tb_gather_indexed: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return launch_gather_indexed(dev, n, d_vec, d_idx, d_out);
}

int tb_scatter_indexed(tb_device *dev, int64_t n, const double *d_in, const int32_t *d_idx, double *d_vec)
{
    TB_REQUIRE(dev && n >= 0 && ((d_vec && d_idx && d_in) || n == 0), "tb_scatter_indexed: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return launch_scatter_indexed(dev, n, d_in, d_idx, d_vec);
}

int tb_scatter_add_indexed(tb_device *dev, int64_t n, const double *d_in, const int32_t *d_idx, double *d_vec)
{
    TB_REQUIRE(dev && n >= 0 && ((d_vec && d_idx && d_in) || n == 0), "tb_scatter_add_indexed: bad argument");
    TB_HIP(hipSetDevice(dev->id));
    return launch_scatter_add_indexed(dev, n, d_in, d_idx, d_vec);
}

int tb_spmv_csr_rows(tb_pattern *pat, const double *d_nzval, const double *d_x, int64_t n_rows, const int32_t *d_rows, double *d_out)
{
    TB_REQUIRE(pat && n_rows >= 0 && ((d_nzval && d_x && d_rows && d_out) || n_rows == 0), "tb_spmv_csr_rows: bad argument");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_spmv_rows(pat, d_nzval, d_x, n_rows, d_rows, d_out);
}

int tb_spmv_csr_dot(tb_pattern *pat, const double *d_nzval, const double *d_x, double *d_y, double *d_dot)
{
    TB_REQUIRE(pat && d_nzval && d_x && d_y && d_dot, "tb_spmv_csr_dot: NULL argument");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_spmv_dot(pat, d_nzval, d_x, d_y, d_dot);
}

int tb_spmv_mirror(tb_pattern *pat, const double *d_nzval)
{
    TB_REQUIRE(pat, "tb_spmv_mirror: pat is NULL");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    if (d_nzval) { std::lock_guard<std::mutex> lock(g_mirrored_mutex); if (std::find(g_mirrored.begin(), g_mirrored.end(), pat) == g_mirrored.end()) g_mirrored.push_back(pat); }
    return launch_mirror_bind(pat, d_nzval);
}

int tb_apply_zero_csr(tb_pattern *pat, double *d_nzval, double *d_f, const uint8_t *d_prescribed, double diag)
{
    TB_REQUIRE(pat && d_prescribed && (d_nzval || d_f), "tb_apply_zero_csr: NULL argument");
    mirror_drop(pat, d_nzval);
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_apply_zero(pat, d_nzval, d_f, d_prescribed, diag);
}

int tb_meandiag(tb_pattern *pat, const double *d_nzval, double *result)
{
    TB_REQUIRE(pat && d_nzval && result, "tb_meandiag: NULL argument");
    TB_HIP(hipSetDevice(pat->mesh->dev->id));
    return launch_meandiag(pat, d_nzval, result);
}

int tb_absmax(tb_device *dev, int64_t n, const double *d_x, int64_t stride, double *result)
{
    TB_REQUIRE(dev && d_x && result && n >= 0 && stride >= 1, "tb_absmax: bad argument");
    if (!n) { *result = 0.0; return TB_OK; }
    return launch_absmax(dev, n, d_x, stride, result);
}

int tb_max(tb_device *dev, int64_t n, const double *d_x, int64_t stride, double *result)
{
    TB_REQUIRE(dev && (d_x || n == 0) && result && n >= 0 && stride >= 1, "tb_max: bad argument");
    return launch_max(dev, n, d_x, stride, result);
}

} // extern "C"
