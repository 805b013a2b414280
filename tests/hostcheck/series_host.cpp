// Host build of the series forms of log / exp / pow that the device closures use (gapflow_amd/csrc/closures.hpp:
// fast_log_series, fast_exp_series, pow_pos_series).  stdin: n, then n values x, n values t, n pairs (x, y), all doubles;
// stdout: log(x)[n], exp(t)[n], pow(x, y)[n].  tests/test_hostcheck.py compares with NumPy, special values included,
// under -fsanitize=address,undefined.
#include <cmath>
#include <cstdio>
#include <vector>

#include "../../gapflow_amd/csrc/closures.hpp"

int main() {
    double nd = 0;
    if (std::fread(&nd, sizeof(double), 1, stdin) != 1) return 2;
    const size_t n = (size_t)nd;
    std::vector<double> in(4 * n), out(3 * n);
    if (std::fread(in.data(), sizeof(double), in.size(), stdin) != in.size()) return 2;
    for (size_t i = 0; i < n; ++i) {
        out[i] = gpf::fast_log_series(in[i]);
        out[n + i] = gpf::fast_exp_series(in[n + i]);
        out[2 * n + i] = gpf::pow_pos_series(in[2 * n + 2 * i], in[2 * n + 2 * i + 1]);
    }
    std::fwrite(out.data(), sizeof(double), out.size(), stdout);
    return 0;
}
