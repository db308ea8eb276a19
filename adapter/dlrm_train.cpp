// C++ host of the Criteo-shaped train step: no Python, no libtorch -- the loop a maintainer of the reference's
// main() (dlrm_libtorch/src/dlrmtest.cpp:415-556) owns, over dlrm_model_* of libdlrm_amd.so.
//
//   dlrm_train [--config mlperf|kaggle] [--batch B (global)] [--steps K] [--warmup W] [--row-cap R]
//              [--world N --rank r --id-file PATH]   one process per GPU; rank 0 writes the communicator id to PATH
//              [--force-exchange]                    one rank, but every exchange of the sharded schedule runs
//              [--row-wise-min-rows R]               tables with at least R rows are split by rows over the ranks
//
// Synthetic batches (uniform ids, N(0,1) dense features, Bernoulli labels: MyDataset of the reference, :299-334,
// with integer ids); every rank draws the ids of its tables for the GLOBAL batch from the batch's seed.
// Prints one JSON line: samples/s over the timed steps (whole job), ms per step, last loss.
#include <hip/hip_runtime_api.h>
#include <unistd.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "../include/dlrm_amd.h"

static void ck(int rc, const char *what) {
    if (rc != DLRM_OK) throw std::runtime_error(std::string(what) + ": " + dlrm_last_error());
}
static void hk(hipError_t e, const char *what) {
    if (e != hipSuccess) throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}
static uint64_t splitmix64(uint64_t x) {
    x += 0x9E3779B97F4A7C15ull;
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

static const int64_t kMlperfRows[26] = {39884406, 39043, 17289, 7420, 20263, 3, 7120, 1543, 63, 38532951, 2953546, 403346, 10,
                                        2208, 11938, 155, 4, 976, 14, 39979771, 25641295, 39664984, 585935, 12972, 108, 36};
static const int64_t kKaggleRows[26] = {1460, 583, 10131227, 2202608, 305, 24, 12517, 633, 3, 93145, 5683, 8351593, 3194, 27,
                                        14992, 5461306, 10, 5652, 2173, 4, 7046547, 18, 15, 286181, 105, 142572};

int main(int argc, char **argv) {
    try {
        std::string config = "mlperf", id_file;
        int64_t batch = 0, row_cap = 0, rw_min = 0;
        int steps = 20, warmup = 3, world = 1, rank = 0, force = 0;
        for (int i = 1; i < argc; i++) {
            std::string a = argv[i];
            auto next = [&]() { if (i + 1 >= argc) throw std::runtime_error("missing value for " + a); return std::string(argv[++i]); };
            if (a == "--config") config = next();
            else if (a == "--batch") batch = std::stoll(next());
            else if (a == "--steps") steps = std::stoi(next());
            else if (a == "--warmup") warmup = std::stoi(next());
            else if (a == "--row-cap") row_cap = std::stoll(next());
            else if (a == "--world") world = std::stoi(next());
            else if (a == "--rank") rank = std::stoi(next());
            else if (a == "--id-file") id_file = next();
            else if (a == "--force-exchange") force = 1;
            else if (a == "--row-wise-min-rows") rw_min = std::stoll(next());
            else throw std::runtime_error("unknown option " + a);
        }
        if (dlrm_version() != DLRM_ABI_VERSION) throw std::runtime_error("libdlrm_amd.so was built from another header revision");
        int n_dev = 0;
        hk(hipGetDeviceCount(&n_dev), "hipGetDeviceCount");
        hk(hipSetDevice(rank % n_dev), "hipSetDevice");

        const bool kaggle = config == "kaggle";
        if (!kaggle && config != "mlperf") throw std::runtime_error("--config mlperf|kaggle");
        const int T = 26, D = kaggle ? 16 : 128;
        std::vector<int64_t> rows(kaggle ? kKaggleRows : kMlperfRows, (kaggle ? kKaggleRows : kMlperfRows) + T);
        if (row_cap) for (auto &r : rows) r = std::min(r, row_cap);
        const std::vector<int32_t> bottom = kaggle ? std::vector<int32_t>{512, 256, 64, 16} : std::vector<int32_t>{512, 256, 128};
        const std::vector<int32_t> top = kaggle ? std::vector<int32_t>{512, 256, 1} : std::vector<int32_t>{1024, 1024, 512, 256, 1};
        if (!batch) batch = kaggle ? 2048 : 65536;
        if (batch % world) throw std::runtime_error("the global batch must divide by the number of ranks");
        const int64_t B = batch / world;

        // ---- communicator: rank 0 draws the id, the others read it from the file
        dlrm_comm *comm = nullptr;
        if (world > 1 || force) {
            uint8_t id[DLRM_COMM_ID_BYTES];
            if (rank == 0) {
                ck(dlrm_comm_unique_id(id), "dlrm_comm_unique_id");
                if (world > 1) {
                    if (id_file.empty()) throw std::runtime_error("--id-file is needed with --world > 1");
                    const std::string tmp = id_file + ".tmp";
                    FILE *f = fopen(tmp.c_str(), "wb");
                    if (!f || fwrite(id, 1, sizeof(id), f) != sizeof(id)) throw std::runtime_error("cannot write " + tmp);
                    fclose(f);
                    if (rename(tmp.c_str(), id_file.c_str())) throw std::runtime_error("cannot rename " + tmp);
                }
            } else {
                FILE *f = nullptr;
                for (int tries = 0; tries < 600 && !(f = fopen(id_file.c_str(), "rb")); tries++) usleep(100000);
                if (!f || fread(id, 1, sizeof(id), f) != sizeof(id)) throw std::runtime_error("cannot read " + id_file);
                fclose(f);
            }
            ck(dlrm_comm_init(id, rank, world, &comm), "dlrm_comm_init");
        }

        // ---- placement: whole tables greedily; tables of at least rw_min rows by row ranges
        std::vector<int32_t> owner;
        if (rw_min > 0) {
            // greedy over the whole tables only (the library's default places all of them whole)
            std::vector<int> order;
            for (int t = 0; t < T; t++) if (rows[t] < rw_min) order.push_back(t);
            std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return rows[a] > rows[b]; });
            std::vector<double> mem(world, 0.0), cnt(world, 0.0);
            owner.assign(T, -1);
            for (int t : order) {
                int best = 0;
                for (int r = 1; r < world; r++)
                    if (cnt[r] < cnt[best] || (cnt[r] == cnt[best] && mem[r] < mem[best])) best = r;
                owner[t] = best; mem[best] += (double)rows[t]; cnt[best] += 1.0;
            }
        }

        dlrm_model_config c{};
        c.num_tables = T; c.table_rows = rows.data(); c.dim = D; c.dense_in = 13;
        c.n_bottom = (int)bottom.size(); c.bottom = bottom.data(); c.n_top = (int)top.size(); c.top = top.data();
        c.interaction = DLRM_INTERACT_DOT; c.loss = DLRM_LOSS_BCE; c.pooling = DLRM_POOL_SUM;
        c.row_dtype = DLRM_F32; c.idx_dtype = DLRM_I32; c.emb_optimizer = DLRM_OPT_SGD; c.gemm_mode = DLRM_GEMM_DEFAULT;
        c.lr = 0.01f; c.eps = 1e-8f; c.local_batch = B; c.rank = rank; c.world = world; c.comm = comm;
        c.table_owner = owner.empty() ? nullptr : owner.data(); c.force_exchange = force;
        c.overlap = -1; c.presort = 1; c.wgrad_stream = -1; c.seed = 0; c.init_tables = 1;
        c.one_id_per_bag = 1;   // the synthetic batches below hold one id per table and sample (offsets[g] = g)
        dlrm_model *m = nullptr;
        ck(dlrm_model_create(&c, &m), "dlrm_model_create");
        ck(dlrm_model_init_dense(m, 0), "dlrm_model_init_dense");
        dlrm_model_info_t info;
        ck(dlrm_model_info(m, &info), "dlrm_model_info");

        // ---- a few synthetic batches, resident in HBM before the timed region
        const int n_batches = 4;
        const int TL = info.num_local_tables;
        std::vector<int32_t> local_global(TL);
        std::vector<int64_t> local_rows(TL);
        for (int i = 0; i < TL; i++) ck(dlrm_model_local_table(m, i, &local_global[i], nullptr, nullptr, nullptr), "local_table");
        for (int i = 0; i < TL; i++) local_rows[i] = rows[local_global[i]];
        const int64_t n_idx = (int64_t)TL * batch;
        std::vector<int32_t *> d_idx(n_batches);
        std::vector<float *> d_dense(n_batches), d_lab(n_batches);
        int32_t *d_off = nullptr;
        {
            std::vector<int32_t> off(n_idx + 1);
            for (int64_t i = 0; i <= n_idx; i++) off[i] = (int32_t)i;
            hk(hipMalloc((void **)&d_off, sizeof(int32_t) * (n_idx + 1)), "hipMalloc");
            hk(hipMemcpy(d_off, off.data(), sizeof(int32_t) * (n_idx + 1), hipMemcpyHostToDevice), "copy");
        }
        for (int b = 0; b < n_batches; b++) {
            std::vector<int32_t> idx(n_idx > 0 ? n_idx : 1);
            for (int i = 0; i < TL; i++) {
                // the ids of a table depend on (batch, table, sample) only: every rank sees the same global batch
                uint64_t s = splitmix64(((uint64_t)b << 40) ^ ((uint64_t)local_global[i] << 32));
                for (int64_t j = 0; j < batch; j++) { s = splitmix64(s); idx[(int64_t)i * batch + j] = (int32_t)(s % (uint64_t)local_rows[i]); }
            }
            std::vector<float> dense((size_t)B * 13), lab(B);
            uint64_t s = splitmix64(0xdeadull + (uint64_t)b * 977 + rank);
            for (auto &x : dense) {
                s = splitmix64(s); const float u1 = (float)(s >> 40) * (1.0f / 16777216.0f) + 1e-7f;
                s = splitmix64(s); const float u2 = (float)(s >> 40) * (1.0f / 16777216.0f);
                x = std::sqrt(-2.0f * std::log(u1)) * std::cos(6.2831853f * u2);
            }
            for (auto &x : lab) { s = splitmix64(s); x = (s >> 40) % 4 == 0 ? 1.0f : 0.0f; }
            hk(hipMalloc((void **)&d_idx[b], sizeof(int32_t) * idx.size()), "hipMalloc");
            hk(hipMalloc((void **)&d_dense[b], sizeof(float) * dense.size()), "hipMalloc");
            hk(hipMalloc((void **)&d_lab[b], sizeof(float) * lab.size()), "hipMalloc");
            hk(hipMemcpy(d_idx[b], idx.data(), sizeof(int32_t) * idx.size(), hipMemcpyHostToDevice), "copy");
            hk(hipMemcpy(d_dense[b], dense.data(), sizeof(float) * dense.size(), hipMemcpyHostToDevice), "copy");
            hk(hipMemcpy(d_lab[b], lab.data(), sizeof(float) * lab.size(), hipMemcpyHostToDevice), "copy");
        }
        hipStream_t st;
        hk(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "stream");
        auto step = [&](int i) {
            const int b = i % n_batches;
            ck(dlrm_model_load_batch(m, d_dense[b], 13, d_lab[b], st), "load_batch");
            ck(dlrm_model_train_step(m, d_idx[b], d_off, n_idx, st), "train_step");
        };
        for (int i = 0; i < warmup; i++) step(i);
        hk(hipStreamSynchronize(st), "sync");
        const auto t0 = std::chrono::steady_clock::now();
        for (int i = 0; i < steps; i++) step(warmup + i);
        hk(hipStreamSynchronize(st), "sync");
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        float loss = 0.f;
        int32_t flag = 0;
        hk(hipMemcpy(&loss, info.d_loss, sizeof(float), hipMemcpyDeviceToHost), "loss");
        hk(hipMemcpy(&flag, info.d_error_flag, sizeof(int32_t), hipMemcpyDeviceToHost), "flag");
        if (flag) throw std::runtime_error(flag == 2 ? "a bag did not hold exactly one id" : "an embedding index was out of range");
        printf("{\"host\": \"c++ (dlrm_model_*)\", \"config\": \"%s\", \"rank\": %d, \"world\": %d, \"global_batch\": %lld, "
               "\"steps\": %d, \"ms_per_step\": %.4f, \"samples_per_s\": %.1f, \"loss_share\": %.6f, \"exchange\": %s}\n",
               config.c_str(), rank, world, (long long)batch, steps, dt * 1e3 / steps, (double)batch * steps / dt, loss,
               (world > 1 || force) ? "true" : "false");
        ck(dlrm_model_destroy(m), "dlrm_model_destroy");
        if (comm) ck(dlrm_comm_destroy(comm), "dlrm_comm_destroy");
        return 0;
    } catch (const std::exception &e) {
        fprintf(stderr, "dlrm_train: %s\n", e.what());
        return 1;
    }
}
