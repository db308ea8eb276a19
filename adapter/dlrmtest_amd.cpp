// Reference-shaped driver: the training loop of dlrm_libtorch/src/dlrmtest.cpp
// (main(), :415-556) on the MI355X path, written against adapter/dlrm_adapter.hpp.
//
//   dlrmtest_amd data_size batch_size cache_mult [options]
//
// The three positional arguments are the reference's (src/dlrmtest.cpp:429-441).
// Model globals that the reference edits in source (:37-49) are options:
//   --tables T --dim D --rows R --lookups L --dense F --epochs E
//   --input FILE     dataset + FC weights (see read_input); default: generated in-process
//   --csv FILE       page traffic log, reference schema 'Time';'Channel';'PU';'Read';'Write' (:448)
//   --trace FILE     per batch and table: lookups that hit / missed, pages read / written / evicted / resident
//                    ('Time';'Epoch';'Batch';'Table';'Hit';'Miss';'Read';'Write';'Evicted';'Resident'; deltas per batch)
//   --canonical      train every table (duplicates accumulate) and the MLP instead of the loop as shipped
//   --policy vec     the cache of src/dlrmtest2.cpp (class cache2): rows cached one by one, cache_mult * lookups *
//                    batch_size listed rows per table (:53), evicted rows repacked into new pages; ids are not sorted.
//                    --slots N (rows of the HBM pool) and --log-pages N (pages a table may write) size it
//   --dump-table0 F  write table 0 (float32 [rows][dim]) after training, for parity checks
//
// As shipped (default) the loop reproduces what the reference executes, not what it
// intends: mean pooling over the batch's SORTED ids, ReLU on all four layers, MSE, and an
// optimizer that only ever holds table 0's rows, written back last-duplicate-wins
// (src/dlrmtest.cpp:79-80, :96-101, :499-506).  One JSON line per batch goes to stdout.
#include <chrono>
#include <cmath>
#include <cstring>
#include <iostream>
#include <memory>

#include "dlrm_adapter.hpp"

using namespace dlrm_amd;

static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}
static float uniform01(uint64_t &s) { s = splitmix64(s); return (float)(s >> 40) * (1.0f / 16777216.0f); }
static float normal01(uint64_t &s) {
    const float u1 = uniform01(s) + 1e-7f, u2 = uniform01(s);
    return std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
}

struct Model {
    int T = 32, D = 32, rows = 500000, L = 80, dense = 256, epochs = 2;   // src/dlrmtest.cpp:37-49 (RMC1)
};

struct Input {
    std::vector<float> data, target;   // [N][dense + T*L], [N]
    std::vector<float> W[4], b[4];
};

// file: int32 magic 0x444c524d, N, T, D, rows, L, dense, 0; float32 data[N][dense+T*L]; target[N];
// then W1[128][dense] b1[128] W2[32][128] b2[32] W3[32][32+T*D] b3[32] W4[1][32] b4[1]
static void read_input(const std::string &path, const Model &m, int N, Input &in) {
    FILE *f = fopen(path.c_str(), "rb");
    if (!f) throw std::runtime_error("cannot open " + path);
    int32_t h[8];
    if (fread(h, 4, 8, f) != 8 || h[0] != 0x444c524d) throw std::runtime_error("bad input header");
    if (h[1] != N || h[2] != m.T || h[3] != m.D || h[4] != m.rows || h[5] != m.L || h[6] != m.dense)
        throw std::runtime_error("input file does not match the model options");
    auto rd = [&](std::vector<float> &v, size_t n) {
        v.resize(n);
        if (fread(v.data(), 4, n, f) != n) throw std::runtime_error("short input file");
    };
    const int cat = 32 + m.T * m.D;
    rd(in.data, (size_t)N * (m.dense + m.T * m.L));
    rd(in.target, N);
    rd(in.W[0], (size_t)128 * m.dense); rd(in.b[0], 128);
    rd(in.W[1], (size_t)32 * 128); rd(in.b[1], 32);
    rd(in.W[2], (size_t)32 * cat); rd(in.b[2], 32);
    rd(in.W[3], 32); rd(in.b[3], 1);
    fclose(f);
}

// MyDataset (src/dlrmtest.cpp:299-334): N(0,1) dense features, uniform ids, N(0,1) labels
static void make_input(const Model &m, int N, Input &in) {
    uint64_t s = 0;
    const int row = m.dense + m.T * m.L, cat = 32 + m.T * m.D;
    in.data.resize((size_t)N * row);
    in.target.resize(N);
    for (int i = 0; i < N; i++) {
        for (int j = 0; j < m.dense; j++) in.data[(size_t)i * row + j] = normal01(s);
        for (int j = m.dense; j < row; j++) in.data[(size_t)i * row + j] = (float)(int)(uniform01(s) * m.rows);
        in.target[i] = normal01(s);
    }
    const int N_[4] = {128, 32, 32, 1}, K_[4] = {m.dense, 128, cat, 32};
    for (int l = 0; l < 4; l++) {
        in.W[l].resize((size_t)N_[l] * K_[l]);
        in.b[l].resize(N_[l]);
        for (auto &w : in.W[l]) w = normal01(s) / std::sqrt((float)K_[l]);
        for (auto &x : in.b[l]) x = std::fabs(normal01(s)) * 0.1f;
    }
}

struct Options {
    Model m;
    int data_size = 800, batch_size = 60, c = 6;   // src/dlrmtest.cpp:45-47
    std::string input, csv_path, dump0, trace_path;
    bool canonical = false, vec = false;
    int64_t slots = 0, log_pages = 0;
};

static cache *make_cache(const Options &o, cache *) {
    const Model &m = o.m;
    const int cache_size = o.c * m.T * m.L * o.batch_size;   // src/dlrmtest.cpp:436
    return new cache(m.T, m.rows, m.D, cache_size, (int64_t)m.T * m.L * o.batch_size + 64);
}
static cache2 *make_cache(const Options &o, cache2 *) {
    const Model &m = o.m;
    const int64_t cache_rows = (int64_t)o.c * m.L * o.batch_size;   // src/dlrmtest2.cpp:509
    const int vpo = 4096 / (m.D * 4);
    // every lookup can bring a whole page; the written region takes what all batches can evict
    const int64_t per_batch = (int64_t)m.L * o.batch_size;
    const int64_t batches = (int64_t)m.epochs * ((o.data_size + o.batch_size - 1) / o.batch_size);
    const int64_t slots = o.slots ? o.slots : std::min<int64_t>((int64_t)m.T * m.rows, (int64_t)m.T * per_batch * vpo * batches);
    const int64_t log_pages = o.log_pages ? o.log_pages : (per_batch * batches + vpo - 1) / vpo + 16;
    return new cache2(m.T, m.rows, m.D, cache_rows, slots, log_pages);
}
static void freeze_capacity(cache &c, int b, int epoch, int mult) {
    if (b + 1 == mult && epoch == 1) c.cache_size = c.page_count;   // src/dlrmtest.cpp:496-498
}
static void freeze_capacity(cache2 &, int, int, int) {}               // dlrmtest2 keeps its capacity
static std::vector<float> table0(cache &c, const Model &m) {
    c.flush();
    const float *t0 = c.store(0);
    return std::vector<float>(t0, t0 + (size_t)m.rows * m.D);
}
static std::vector<float> table0(cache2 &c, const Model &m) { return c.read_rows(0, m.rows); }

template <class Cache>
static int train(Options o);

int main(int argc, char *argv[]) {
    try {
        Options o;
        Model &m = o.m;
        int &data_size = o.data_size, &batch_size = o.batch_size, &c = o.c;
        std::string &input = o.input, &csv_path = o.csv_path, &dump0 = o.dump0, &trace_path = o.trace_path;
        bool &canonical = o.canonical;
        std::vector<std::string> pos;
        for (int i = 1; i < argc; i++) {
            std::string a = argv[i];
            auto next = [&]() { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); return std::string(argv[++i]); };
            if (a == "--tables") m.T = std::stoi(next());
            else if (a == "--dim") m.D = std::stoi(next());
            else if (a == "--rows") m.rows = std::stoi(next());
            else if (a == "--lookups") m.L = std::stoi(next());
            else if (a == "--dense") m.dense = std::stoi(next());
            else if (a == "--epochs") m.epochs = std::stoi(next());
            else if (a == "--input") input = next();
            else if (a == "--csv") csv_path = next();
            else if (a == "--trace") trace_path = next();
            else if (a == "--canonical") canonical = true;
            else if (a == "--dump-table0") dump0 = next();
            else if (a == "--policy") { const std::string v = next(); if (v != "vec" && v != "page") throw std::runtime_error("--policy page|vec"); o.vec = v == "vec"; }
            else if (a == "--slots") o.slots = std::stoll(next());
            else if (a == "--log-pages") o.log_pages = std::stoll(next());
            else pos.push_back(a);
        }
        if (pos.size() == 3) { data_size = std::stoi(pos[0]); batch_size = std::stoi(pos[1]); c = std::stoi(pos[2]); }
        else if (!pos.empty()) throw std::runtime_error("usage: dlrmtest_amd data_size batch_size cache_mult [options]");
        return o.vec ? train<cache2>(o) : train<cache>(o);
    } catch (const std::exception &e) {
        std::cerr << "dlrmtest_amd: " << e.what() << std::endl;
        return 1;
    }
}

template <class Cache>
static int train(Options o) {
    {
        Model &m = o.m;
        int data_size = o.data_size, batch_size = o.batch_size, c = o.c;
        const std::string &input = o.input, &csv_path = o.csv_path, &dump0 = o.dump0, &trace_path = o.trace_path;
        const bool canonical = o.canonical;
        constexpr int CS = Cache::counter_stride;

        std::string name = csv_path.empty() ? std::string(o.vec ? "LRU_AMD_C" : "LRU_BASE_AMD_C") + std::to_string(c) + "_B" + std::to_string(batch_size) + "_D" +
                                                  std::to_string(data_size) + ".csv" : csv_path;
        std::ofstream myfile(name);
        myfile << "'Time';'Channel';'PU';'Read';'Write'" << std::endl;   // :448

        std::ofstream trace;
        std::vector<int64_t> trace_prev;
        if (!trace_path.empty()) {
            trace.open(trace_path);
            trace << "'Time';'Epoch';'Batch';'Table';'Hit';'Miss';'Read';'Write';'Evicted';'Resident'" << std::endl;
        }

        Input in;
        if (input.empty()) make_input(m, data_size, in); else read_input(input, m, data_size, in);

        const auto t_start = std::chrono::steady_clock::now();
        auto now = [&]() { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count(); };

        Net2 net2(m.dense, m.T, m.D);
        for (int l = 0; l < 4; l++) net2.set_layer(l, in.W[l], in.b[l]);
        std::unique_ptr<Cache> cache_holder(make_cache(o, (Cache *)nullptr));
        Cache &emb_cache = *cache_holder;
        emb_cache.write();

        const int row = m.dense + m.T * m.L, cat = net2.cat, bb = batch_size;
        DeviceArray<float> d_x((size_t)bb * m.dense), d_t(bb), dpred(bb), dz3((size_t)bb * 32), dcat((size_t)bb * cat),
            dz2((size_t)bb * 32), dz1((size_t)bb * 128), loss(1);
        DeviceArray<float> &h1 = net2.h1, &catbuf = net2.catbuf, &h3 = net2.h3, &pred = net2.pred;
        DeviceArray<float> dW[4], db[4];
        DeviceArray<char> ws, ws_dense;
        const int N_[4] = {128, 32, 32, 1}, K_[4] = {m.dense, 128, cat, 32};
        if (canonical)
            for (int l = 0; l < 4; l++) { dW[l].resize((size_t)N_[l] * K_[l]); db[l].resize(N_[l]); }

        for (int epoch = 1; epoch <= m.epochs; ++epoch) {
            int b = 0;
            for (int s0 = 0; s0 < data_size; s0 += bb, b++) {
                batch_size = std::min(bb, data_size - s0);   // last partial batch (:479-482)
                const int B = batch_size;
                const float *xb = in.data.data() + (size_t)s0 * row;
                emb_t<Cache> embedding(xb, B, m.dense, m.L, emb_cache, &myfile, now());
                // The page cache reclaims here, like the reference (:494): the pages of the batch are its most
                // recent ones and stay.  The vector cache lists rows by their FIRST request, so a reclaim may drop
                // rows of this very batch (the reference works on a copy and loses their update,
                // src/dlrmtest2.cpp:558-559, :595): it runs after the rows are updated instead -- the cache sees
                // the same calls in the same order either way.
                auto reclaim_now = [&]() {
                    reclaim_emb_cache(emb_cache, &myfile, now());
                    freeze_capacity(emb_cache, b, epoch, c);
                    embedding.restore();
                };
                if (Cache::sorts_ids) reclaim_now();
                auto write_trace = [&]() {
                if (trace.is_open()) {
                    const std::vector<int64_t> cur = emb_cache.counters();
                    if (trace_prev.empty()) trace_prev.assign(cur.size(), 0);
                    const double tnow = now();
                    for (int t = 0; t < m.T; t++) {
                        trace << tnow << ";" << epoch << ";" << b << ";" << t;
                        static const int order[5] = {2, 3, 0, 1, 4};   // hit, miss, read, write, evicted
                        for (int k : order) trace << ";" << cur[t * CS + k] - trace_prev[t * CS + k];
                        trace << ";" << cur[t * CS + 5] << "\n";
                    }
                    trace_prev = cur;
                }
                };
                if (Cache::sorts_ids) write_trace();

                std::vector<float> xd((size_t)B * m.dense);
                for (int i = 0; i < B; i++) std::memcpy(&xd[(size_t)i * m.dense], xb + (size_t)i * row, sizeof(float) * m.dense);
                d_x.upload(xd);
                d_t.upload(in.target.data() + s0, B);

                // net2->forward(dense, embedding->forward(lookup)) (:514-518, :254-272)
                net2.forward(d_x.p, B, [&](float *catp, int64_t ld, int col0) { embedding.forward(catp, ld, col0); });
                check(dlrm_loss_mse(pred.p, d_t.p, B, B, loss.p, dpred.p, 1, nullptr), "dlrm_loss_mse");   // :522

                // loss.backward() (:531) down to the pooled embeddings
                dlrm_linear_bwd l4{};
                l4.M = B; l4.N = 1; l4.K = 32; l4.d_dz = dpred.p; l4.dz_ld = 1; l4.d_w = net2.W[3].p; l4.w_ld = 32;
                l4.d_dx = dz3.p; l4.dx_ld = 32; l4.d_relu_mask = h3.p; l4.relu_mask_ld = 32;
                check(dlrm_linear_backward_data(&l4, nullptr), "bwd fc4");
                dlrm_linear_bwd l3{};
                l3.M = B; l3.N = 32; l3.K = cat; l3.d_dz = dz3.p; l3.dz_ld = 32; l3.d_w = net2.W[2].p; l3.w_ld = cat;
                l3.d_dx = dcat.p; l3.dx_ld = cat;
                check(dlrm_linear_backward_data(&l3, nullptr), "bwd fc3");

                if (!canonical) {
                    // the optimizer holds embedding->parameters()[0] only (:502-506): table 0
                    embedding.update_parameter(0, dcat.p + 32, cat, 0.01f, true, ws);
                } else {
                    for (int t = 0; t < m.T; t++) embedding.update_parameter(t, dcat.p + 32 + (size_t)t * m.D, cat, 0.01f, false, ws);
                    // dense layers: gradients + SGD
                    auto wgrad = [&](int l, const float *dz, int64_t dz_ld, const float *x, int64_t x_ld) {
                        dlrm_linear_bwd g{};
                        g.M = B; g.N = N_[l]; g.K = K_[l]; g.d_dz = dz; g.dz_ld = dz_ld; g.d_x = x; g.x_ld = x_ld;
                        g.d_dw = dW[l].p; g.dw_ld = K_[l]; g.d_db = db[l].p;
                        size_t need = 0;
                        check(dlrm_linear_backward_weight_workspace_bytes(B, N_[l], K_[l], &need), "ws");
                        ws_dense.resize(need);
                        check(dlrm_linear_backward_weight(&g, ws_dense.p, ws_dense.n, nullptr), "wgrad");
                    };
                    wgrad(3, dpred.p, 1, h3.p, 32);
                    wgrad(2, dz3.p, 32, catbuf.p, cat);
                    // bottom: dz2 = (dz3 W3[:, :32]) masked by the ReLU of fc2's output (= catbuf[:, :32])
                    dlrm_linear_bwd l3b{};
                    l3b.M = B; l3b.N = 32; l3b.K = 32; l3b.d_dz = dz3.p; l3b.dz_ld = 32; l3b.d_w = net2.W[2].p; l3b.w_ld = cat;
                    l3b.d_dx = dz2.p; l3b.dx_ld = 32; l3b.d_relu_mask = catbuf.p; l3b.relu_mask_ld = cat;
                    check(dlrm_linear_backward_data(&l3b, nullptr), "bwd fc3 (bottom part)");
                    wgrad(1, dz2.p, 32, h1.p, 128);
                    dlrm_linear_bwd l2{};
                    l2.M = B; l2.N = 32; l2.K = 128; l2.d_dz = dz2.p; l2.dz_ld = 32; l2.d_w = net2.W[1].p; l2.w_ld = 128;
                    l2.d_dx = dz1.p; l2.dx_ld = 128; l2.d_relu_mask = h1.p; l2.relu_mask_ld = 128;
                    check(dlrm_linear_backward_data(&l2, nullptr), "bwd fc2");
                    wgrad(0, dz1.p, 128, d_x.p, m.dense);
                    for (int l = 0; l < 4; l++) {
                        check(dlrm_sgd_step(net2.W[l].p, dW[l].p, (int64_t)N_[l] * K_[l], 0.01f, nullptr), "sgd W");
                        check(dlrm_sgd_step(net2.b[l].p, db[l].p, N_[l], 0.01f, nullptr), "sgd b");
                    }
                }
                if (!Cache::sorts_ids) { reclaim_now(); write_trace(); }
                const float lv = loss.download(1)[0];
                std::cout << "{\"epoch\": " << epoch << ", \"batch\": " << b << ", \"batch_size\": " << B << ", \"loss\": " << lv
                          << ", \"page_count\": " << emb_cache.page_count << ", \"cache_size\": " << emb_cache.cache_size << "}"
                          << std::endl;
            }
        }
        const int64_t flags = emb_cache.error_flags();
        // checksum of table 0 after training (for parity checks) and timing like the reference (:548-554)
        double sum0 = 0.0;
        const std::vector<float> t0 = table0(emb_cache, m);
        for (size_t i = 0; i < t0.size(); i++) sum0 += t0[i];
        if (!dump0.empty()) {
            FILE *f = fopen(dump0.c_str(), "wb");
            if (!f || fwrite(t0.data(), 4, t0.size(), f) != t0.size()) throw std::runtime_error("cannot write " + dump0);
            fclose(f);
        }
        std::cout << "{\"table0_sum\": " << sum0 << ", \"error_flags\": " << flags << ", \"training_time_s\": " << now() << "}" << std::endl;
        emb_cache.dump_chunk_sector_stats("chunk_statistics");
        return 0;
    }
}
