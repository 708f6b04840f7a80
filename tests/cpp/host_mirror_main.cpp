// A C++ program against the host mirror alone (no Python): the CLI run of the reference's
// test.cpp:27-40 (MyState, Target, MetropolisChain(t, 0).run(...)).  On a machine without a
// GPU every device call must fail loudly: exit code 3 and the C ABI's message.
#include <cstdio>
#include "mcmc_host.hpp"

int main(int argc, char** argv) {
    using namespace mcmc_amd;
    const int nSamples = argc > 1 ? std::atoi(argv[1]) : 100000;
    try {
        auto t = std::make_shared<Target>();
        auto s = std::make_shared<MyState>();                 // init() evaluates on the device
        t->set_posterior(s);
        MetropolisChain chain(t, 0);
        chain.computeMean = true;                             // false by default, mcmc.h:990
        chain.run(nSamples, 10, 500, 5);                      // test.cpp:38
        size_t len = 0;
        std::vector<Float> pos = chain.get_samples("position", &len);
        std::vector<Float> mean = chain.get_mean("position");
        std::printf("saved %zu x %zu, mean %.17g %.17g %.17g, last loglike %.17g\n", pos.size() / len, len, mean[0], mean[1],
                    mean[2], chain.get_loglikes().back());
        return 0;
    } catch (const HipError& e) {
        std::printf("HipError %d: %s\n", e.code, e.what());
        return 3;
    }
}
