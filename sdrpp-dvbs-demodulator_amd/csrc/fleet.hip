// The multi-GPU split BEHIND the boundary (include/dvbs2gpu.h, dvbs2gpu_fleet_*): what a C++ plugin host with several GPUs calls.
//
// The reference's unit of parallelism is one independent DVBS2Demod instance per transponder (src/main.cpp:588,595); nothing is exchanged
// inside a frame or between streams.  A fleet is a set of MEMBERS -- one engine context + one worker thread per (logical) device -- and a
// transponder table placed on them by the rule of sdrpp-dvbs-demodulator_amd/distribute.py (assign_transponders: whole MODCOD groups by
// longest-processing-time first, so that every member's LDPC batches stay homogeneous, else the MODCOD-sorted list cut into pieces of
// near-equal weight).  dvbs2gpu_fleet_process_batch hands every member its transponders' samples (host buffers, as a plugin host has
// them), the members run dvbs2gpu_demod_process_batch side by side, and the BBFRAMEs come back in TRANSPONDER (table) order -- the egress
// order of the plugin's sink, main.cpp:532-558.  No data-path collective: the devices never talk to each other (the RCCL broadcast /
// gather of the multi-process harness, distribute.py, is work distribution between PROCESSES; here one process owns all devices).
// Host-side only; no exception crosses the ABI; a member's error comes back as the call's return code with its message in
// dvbs2gpu_last_error().
#include "ctx.h"

#include <algorithm>
#include <condition_variable>
#include <map>
#include <memory>
#include <thread>

using namespace s2;

namespace {

// distribute.py: shard_by_weight
static std::vector<std::vector<int>> lpt(const std::vector<double>& w, int world) {
    std::vector<int> order(w.size());
    for (size_t i = 0; i < w.size(); ++i) order[i] = (int)i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return w[a] > w[b]; });
    std::vector<double> loads(world, 0.0);
    std::vector<std::vector<int>> out(world);
    for (int i : order) {
        int r = 0;
        for (int k = 1; k < world; ++k) if (loads[k] < loads[r]) r = k;
        out[r].push_back(i);
        loads[r] += w[i];
    }
    return out;
}

// distribute.py: assign_transponders -- same arithmetic in the same order, so that both give the same placement for the same table
static std::vector<std::vector<int>> place(const int32_t* modcod, const double* weight, int n, int world, double tolerance) {
    std::vector<std::vector<int>> out(std::max(world, 1));
    if (world <= 1 || n == 0) {
        for (int i = 0; i < n; ++i) out[0].push_back(i);
        return out;
    }
    std::map<int, std::vector<int>> groups;          // (ascending MODCOD, members in table order)
    for (int i = 0; i < n; ++i) groups[modcod[i]].push_back(i);
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += weight[i];
    if (total == 0.0) total = 1.0;
    const double avg = total / world;
    auto gsum = [&](const std::vector<int>& g) { double s = 0.0; for (int i : g) s += weight[i]; return s; };
    std::vector<int> keys;
    for (auto& kv : groups) keys.push_back(kv.first);
    std::stable_sort(keys.begin(), keys.end(), [&](int a, int b) { const double sa = gsum(groups[a]), sb = gsum(groups[b]); return sa != sb ? sa > sb : a < b; });
    if ((int)keys.size() >= world) {
        std::vector<double> gw;
        for (int m : keys) gw.push_back(gsum(groups[m]));
        const auto placed = lpt(gw, world);
        std::vector<std::vector<int>> cand(world);
        double worst = 0.0;
        for (int r = 0; r < world; ++r) {
            for (int g : placed[r]) for (int i : groups[keys[g]]) cand[r].push_back(i);
            std::sort(cand[r].begin(), cand[r].end());
            worst = std::max(worst, gsum(cand[r]));
        }
        if (worst <= (1.0 + tolerance) * avg) return cand;
    }
    std::vector<int> order;
    for (auto& kv : groups) for (int i : kv.second) order.push_back(i);
    double acc = 0.0;
    int r = 0;
    for (size_t pos = 0; pos < order.size(); ++pos) {
        const int i = order[pos];
        const int left = (int)(order.size() - pos);
        if (r < world - 1 && !out[r].empty() && (acc + 0.5 * weight[i] > (r + 1) * avg || left <= world - 1 - r)) ++r;
        out[r].push_back(i);
        acc += weight[i];
    }
    for (auto& x : out) std::sort(x.begin(), x.end());
    return out;
}

struct Unit {                 // one transponder on its member
    int table_index = -1;
    dvbs2gpu_demod* h = nullptr;
    float* d_iq = nullptr;    // [max_samples] complex64
    uint8_t* d_out = nullptr; // [out_cap]
    int max_samples = 0;
};

struct Member {
    int device = 0;
    dvbs2gpu_ctx* ctx = nullptr;
    std::vector<Unit> units;
    // worker thread + its mailbox
    std::thread th;
    std::mutex m;
    std::condition_variable cv;
    bool quit = false;
    unsigned long long posted = 0, done = 0;
    // the job
    const float* const* h_iq = nullptr;
    const int* counts = nullptr;
    uint8_t* const* h_out = nullptr;
    int out_cap = 0;
    int* out_bytes = nullptr;
    int rc = 0;
    std::string err;
};

}  // namespace

struct dvbs2gpu_fleet {
    std::vector<std::unique_ptr<Member>> members;
    std::vector<int> member_of;       // [nt]
    int nt = 0, out_cap = 0;
    std::mutex call_mtx;
};

namespace {

static void run_job(Member& M) {
    M.rc = 0; M.err.clear();
    auto fail = [&](int rc, const std::string& what) { M.rc = rc; M.err = "fleet member on device " + std::to_string(M.device) + ": " + what; };
    try {
        if (hipSetDevice(M.device) != hipSuccess) return fail(DVBS2GPU_ERR_HIP, "hipSetDevice");
        const int n = (int)M.units.size();
        std::vector<dvbs2gpu_demod*> hs(n);
        std::vector<const float*> iq(n);
        std::vector<int> cnt(n), nb(n, 0);
        std::vector<uint8_t*> outs(n);
        for (int i = 0; i < n; ++i) {
            const Unit& U = M.units[i];
            const int c = M.counts[U.table_index];
            if (c < 0 || c > U.max_samples) return fail(DVBS2GPU_ERR_ARG, "more samples than the transponder's max_samples");
            if (c > 0 && hipMemcpy(U.d_iq, M.h_iq[U.table_index], sizeof(float) * 2 * (size_t)c, hipMemcpyHostToDevice) != hipSuccess) return fail(DVBS2GPU_ERR_HIP, "hipMemcpy(samples)");
            hs[i] = U.h; iq[i] = U.d_iq; cnt[i] = c; outs[i] = U.d_out;
        }
        const int rc = dvbs2gpu_demod_process_batch(hs.data(), n, iq.data(), cnt.data(), outs.data(), M.out_cap, nb.data());
        if (rc != 0) return fail(rc, dvbs2gpu_last_error());
        for (int i = 0; i < n; ++i) {
            const Unit& U = M.units[i];
            if (nb[i] > 0 && hipMemcpy(M.h_out[U.table_index], U.d_out, (size_t)nb[i], hipMemcpyDeviceToHost) != hipSuccess) return fail(DVBS2GPU_ERR_HIP, "hipMemcpy(BBFRAMEs)");
            M.out_bytes[U.table_index] = nb[i];
        }
    } catch (const std::exception& e) {
        fail(DVBS2GPU_ERR_ARG, e.what());
    } catch (...) {
        fail(DVBS2GPU_ERR_ARG, "unknown exception");
    }
}

static void worker(Member* M) {
    std::unique_lock<std::mutex> l(M->m);
    for (;;) {
        M->cv.wait(l, [&] { return M->quit || M->posted != M->done; });
        if (M->quit) return;
        l.unlock();
        run_job(*M);
        l.lock();
        M->done = M->posted;
        M->cv.notify_all();
    }
}

static void release_units(Member& M) {
    (void)hipSetDevice(M.device);
    for (Unit& U : M.units) {
        if (U.h) dvbs2gpu_demod_destroy(U.h);
        (void)hipFree(U.d_iq); (void)hipFree(U.d_out);
    }
    M.units.clear();
}

}  // namespace

extern "C" {

int dvbs2gpu_fleet_plan(const int32_t* modcods, const double* weights, int nt, int world, double tolerance, int32_t* member_of) {
    if (nt < 0 || world < 1 || (nt > 0 && (!modcods || !weights || !member_of))) return DVBS2GPU_ERR_ARG;
    try {
        const auto out = place(modcods, weights, nt, world, tolerance);
        for (int r = 0; r < (int)out.size(); ++r) for (int i : out[r]) member_of[i] = r;
    } catch (...) { last_error() = "fleet plan: out of memory"; return DVBS2GPU_ERR_ARG; }
    return DVBS2GPU_OK;
}

int dvbs2gpu_fleet_create(const int* devices, int n, dvbs2gpu_fleet** out) {
    if (!devices || n < 1 || !out) return DVBS2GPU_ERR_ARG;
    *out = nullptr;
    try {
        std::unique_ptr<dvbs2gpu_fleet> f(new dvbs2gpu_fleet());
        for (int i = 0; i < n; ++i) {
            std::unique_ptr<Member> M(new Member());
            M->device = devices[i];
            const int rc = dvbs2gpu_create(devices[i], &M->ctx);
            if (rc != 0) {
                for (auto& P : f->members) dvbs2gpu_destroy(P->ctx);
                return rc;
            }
            f->members.push_back(std::move(M));
        }
        dvbs2gpu_fleet* raw = f.release();          // (from here on dvbs2gpu_fleet_destroy cleans up: it joins whatever threads have started)
        try {
            for (auto& M : raw->members) M->th = std::thread(worker, M.get());
        } catch (const std::exception& e) {
            dvbs2gpu_fleet_destroy(raw);
            last_error() = std::string("fleet create: worker thread: ") + e.what();
            return DVBS2GPU_ERR_ARG;
        }
        *out = raw;
    } catch (const std::exception& e) { last_error() = std::string("fleet create: ") + e.what(); return DVBS2GPU_ERR_ARG; }
    return DVBS2GPU_OK;
}

void dvbs2gpu_fleet_destroy(dvbs2gpu_fleet* f) {
    if (!f) return;
    { std::lock_guard<std::mutex> guard(f->call_mtx); }          // (a call still running on another thread finishes first; destroying a fleet while calling into it is the host's bug)
    for (auto& M : f->members) {
        { std::lock_guard<std::mutex> l(M->m); M->quit = true; }
        M->cv.notify_all();
        if (M->th.joinable()) M->th.join();
        release_units(*M);
        dvbs2gpu_destroy(M->ctx);
    }
    delete f;
}

int dvbs2gpu_fleet_size(const dvbs2gpu_fleet* f) { return f ? (int)f->members.size() : DVBS2GPU_ERR_ARG; }

int dvbs2gpu_fleet_assign(dvbs2gpu_fleet* f, const dvbs2gpu_fleet_entry* table, int nt, int out_cap, double tolerance, int32_t* member_of) {
    if (!f || nt < 0 || (nt > 0 && !table) || out_cap < 0) return DVBS2GPU_ERR_ARG;
    std::lock_guard<std::mutex> guard(f->call_mtx);
    try {
        for (auto& M : f->members) release_units(*M);
        f->member_of.assign(nt, 0); f->nt = nt; f->out_cap = out_cap;
        std::vector<int32_t> mod(nt);
        std::vector<double> w(nt);
        for (int i = 0; i < nt; ++i) {
            const dvbs2gpu_demod_cfg& c = table[i].cfg;
            mod[i] = c.modcod;
            w[i] = table[i].weight;
            if (!(w[i] > 0.0)) {
                // default weight: LDPC edge visits per frame + a front-end term per symbol (the weighting bench.py's mixed batches use)
                dvbs2gpu_modcod_info mi;
                const int rc = dvbs2gpu_modcod_info_get(c.modcod, c.shortframes, c.pilots, &mi);
                if (rc != 0) return rc;
                const int iters = c.force_ldpc_iters > 0 ? c.force_ldpc_iters : c.max_ldpc_trials;
                w[i] = (double)mi.ldpc_edges * iters + 40.0 * mi.plframe_symbols;
            }
        }
        const auto placed = place(mod.data(), w.data(), nt, (int)f->members.size(), tolerance);
        for (int r = 0; r < (int)placed.size(); ++r) {
            Member& M = *f->members[r];
            if (!placed[r].empty() && hipSetDevice(M.device) != hipSuccess) { last_error() = "fleet assign: hipSetDevice"; return DVBS2GPU_ERR_HIP; }
            for (int i : placed[r]) {
                f->member_of[i] = r;
                Unit U;
                U.table_index = i; U.max_samples = table[i].max_samples;
                int rc = dvbs2gpu_demod_create(M.ctx, &table[i].cfg, table[i].max_samples, &U.h);
                if (rc == 0 && hipMalloc((void**)&U.d_iq, sizeof(float) * 2 * (size_t)std::max(table[i].max_samples, 1)) != hipSuccess) rc = fail_hip(hipGetLastError(), "hipMalloc(fleet samples)");
                if (rc == 0 && hipMalloc((void**)&U.d_out, (size_t)std::max(out_cap, 1)) != hipSuccess) rc = fail_hip(hipGetLastError(), "hipMalloc(fleet output)");
                M.units.push_back(U);          // (kept even on failure: release_units gives back what was made)
                if (rc != 0) {
                    const std::string msg = last_error();
                    for (auto& P : f->members) release_units(*P);
                    f->nt = 0;
                    last_error() = msg;
                    return rc;
                }
            }
        }
        if (member_of) for (int i = 0; i < nt; ++i) member_of[i] = f->member_of[i];
    } catch (const std::exception& e) { last_error() = std::string("fleet assign: ") + e.what(); return DVBS2GPU_ERR_ARG; }
    return DVBS2GPU_OK;
}

int dvbs2gpu_fleet_set_pipelined(dvbs2gpu_fleet* f, int on) {
    if (!f) return DVBS2GPU_ERR_ARG;
    std::lock_guard<std::mutex> guard(f->call_mtx);
    for (auto& M : f->members) { const int rc = dvbs2gpu_set_pipelined(M->ctx, on); if (rc != 0) return rc; }
    return DVBS2GPU_OK;
}

int dvbs2gpu_fleet_reset(dvbs2gpu_fleet* f) {
    if (!f) return DVBS2GPU_ERR_ARG;
    std::lock_guard<std::mutex> guard(f->call_mtx);
    for (auto& M : f->members) for (Unit& U : M->units) { const int rc = dvbs2gpu_demod_reset(U.h); if (rc != 0) return rc; }
    return DVBS2GPU_OK;
}

int dvbs2gpu_fleet_process_batch(dvbs2gpu_fleet* f, const float* const* h_iq, const int* counts, uint8_t* const* h_out, int out_cap, int* out_bytes) {
    if (!f || (f->nt > 0 && (!h_iq || !counts || !h_out || !out_bytes)) || out_cap < 0 || out_cap > f->out_cap) {
        last_error() = "fleet process_batch: bad arguments (out_cap beyond the capacity given to dvbs2gpu_fleet_assign?)";
        return DVBS2GPU_ERR_ARG;
    }
    std::lock_guard<std::mutex> guard(f->call_mtx);
    for (int i = 0; i < f->nt; ++i) out_bytes[i] = 0;
    for (auto& M : f->members) {
        if (M->units.empty()) continue;
        std::lock_guard<std::mutex> l(M->m);
        M->h_iq = h_iq; M->counts = counts; M->h_out = h_out; M->out_cap = out_cap; M->out_bytes = out_bytes;
        ++M->posted;
        M->cv.notify_all();
    }
    int rc = DVBS2GPU_OK;
    for (auto& M : f->members) {
        if (M->units.empty()) continue;
        std::unique_lock<std::mutex> l(M->m);
        M->cv.wait(l, [&] { return M->posted == M->done; });
        if (M->rc != 0 && rc == DVBS2GPU_OK) { rc = M->rc; last_error() = M->err; }      // (every member is waited for, the first error is reported)
    }
    return rc;
}

int dvbs2gpu_fleet_get_stats(dvbs2gpu_fleet* f, int transponder, dvbs2gpu_frame_stats* h_out, int cap) {
    if (!f || transponder < 0 || transponder >= f->nt) return DVBS2GPU_ERR_ARG;
    std::lock_guard<std::mutex> guard(f->call_mtx);
    for (Unit& U : f->members[f->member_of[transponder]]->units)
        if (U.table_index == transponder) return dvbs2gpu_demod_get_stats(U.h, h_out, cap);
    return DVBS2GPU_ERR_ARG;
}

}  // extern "C"
