// Test infrastructure only (nothing under dlrm_amd/ links or runs this).
//
// The reference's own toy driver, dlrm_libtorch/dlrmtest.cpp, compiled from where it lies (REF_SRC, given on the
// command line by oracle/Makefile; the source is included, never copied) with its main() renamed, behind a main()
// that runs ONE batch the way the reference's loop does (dlrmtest.cpp:339-372: make_shared<Net2>(data), forward,
// mse_loss, backward) and prints every tensor involved as JSON: the batch, the layer weights the reference binds
// (its globals fc1_weight .. fc4_weight), the biases of its torch::nn::Linear modules, the two EmbeddingBag tables
// its create_emb builds, the prediction, the loss and all gradients.  tests/golden/reference_net2_step.json is
// this program's output; tests/test_oracle_reference_net2.py holds the CPU oracle to it.
//
// What is set from here: the VALUES of the reference's global weight tensors (it draws them N(0,1) at static
// initialisation, before any seed can be set; unscaled they saturate the final ReLU) and the seed of the batch.
// Everything that computes is the reference's code and this container's libtorch.
#define main reference_main
#include REF_SRC
#undef main

#include <cstdio>
#include <sstream>

static void dump(const char *name, const torch::Tensor &t_in, bool last = false) {
    torch::Tensor t = t_in.detach().to(torch::kFloat32).contiguous();
    std::printf("  \"%s\": {\"shape\": [", name);
    for (int64_t i = 0; i < t.dim(); i++) std::printf("%s%lld", i ? ", " : "", (long long)t.size(i));
    std::printf("], \"data\": [");
    const float *p = t.data_ptr<float>();
    for (int64_t i = 0; i < t.numel(); i++) std::printf("%s%.9g", i ? ", " : "", p[i]);
    std::printf("]}%s\n", last ? "" : ",");
}

int main() {
    // the prediction must be positive for both samples, or the final ReLU stops every gradient: first seed that does
    for (int seed = 1; seed < 200; seed++) {
        torch::manual_seed(seed);
        {
            torch::NoGradGuard ng;
            fc1_weight.copy_(torch::randn_like(fc1_weight) / std::sqrt((double)fc1_weight.size(1)));
            fc2_weight.copy_(torch::randn_like(fc2_weight) / std::sqrt((double)fc2_weight.size(1)));
            fc3_weight.copy_(torch::randn_like(fc3_weight) / std::sqrt((double)fc3_weight.size(1)));
            fc4_weight.copy_(torch::abs(torch::randn_like(fc4_weight)) / std::sqrt((double)fc4_weight.size(1)));
            for (auto *w : {&fc1_weight, &fc2_weight, &fc3_weight, &fc4_weight})
                if (w->grad().defined()) w->mutable_grad() = torch::Tensor();
        }
        torch::Tensor data = torch::cat({torch::randn({batch_size, dense_feature}),
                                         torch::randint(emb_size, {batch_size, emb_table * lookup_num}).to(torch::kFloat32)}, 1);
        torch::Tensor target = torch::randn({batch_size, 1});
        std::ostringstream sink;                               // (the reference prints shapes while it runs)
        std::streambuf *console = std::cout.rdbuf(sink.rdbuf());
        auto net2 = std::make_shared<Net2>(data);
        torch::Tensor prediction = net2->forward(data);
        std::cout.rdbuf(console);
        if (!(prediction > 0).all().item<bool>()) continue;
        torch::Tensor loss = torch::mse_loss(prediction, target);
        loss.backward();
        std::printf("{\n  \"source\": \"dlrm_libtorch/dlrmtest.cpp: struct Net2 (:129-240), create_emb (:20-52), create_fc (:85-105), "
                    "loop body (:339-372); libtorch %s\",\n  \"seed\": %d,\n", TORCH_VERSION, seed);
        dump("data", data);
        dump("target", target);
        dump("fc1_weight", fc1_weight); dump("fc2_weight", fc2_weight); dump("fc3_weight", fc3_weight); dump("fc4_weight", fc4_weight);
        dump("fc1_bias", net2->fc1->bias); dump("fc2_bias", net2->fc2->bias); dump("fc3_bias", net2->fc3->bias); dump("fc4_bias", net2->fc4->bias);
        dump("emb0_weight", net2->emb0->weight); dump("emb1_weight", net2->emb1->weight);
        dump("prediction", prediction);
        dump("loss", loss.reshape({1}));
        dump("fc1_weight_grad", fc1_weight.grad()); dump("fc2_weight_grad", fc2_weight.grad());
        dump("fc3_weight_grad", fc3_weight.grad()); dump("fc4_weight_grad", fc4_weight.grad());
        dump("fc1_bias_grad", net2->fc1->bias.grad()); dump("fc2_bias_grad", net2->fc2->bias.grad());
        dump("fc3_bias_grad", net2->fc3->bias.grad()); dump("fc4_bias_grad", net2->fc4->bias.grad());
        dump("emb0_weight_grad", net2->emb0->weight.grad()); dump("emb1_weight_grad", net2->emb1->weight.grad(), true);
        std::printf("}\n");
        return 0;
    }
    std::fprintf(stderr, "no seed gave a positive prediction for every sample\n");
    return 1;
}
