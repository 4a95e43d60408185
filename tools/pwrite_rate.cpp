// Page-cache write rates of the box: N files x T threads each, pwrite of 8 MB pieces at disjoint offsets (what the table writers do).
//   g++ -O2 -pthread tools/pwrite_rate.cpp -o /tmp/pwrite_rate && /tmp/pwrite_rate DIR
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <string>
#include <thread>
#include <unistd.h>
#include <vector>
#include <atomic>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char** argv) {
    const std::string dir = argc > 1 ? argv[1] : "/tmp";
    const size_t piece = 8u << 20;
    std::vector<char> buf(piece); for (size_t i = 0; i < piece; ++i) buf[i] = (char)('a' + i % 23);
    const size_t per_file = (size_t)((argc > 2 ? atof(argv[2]) : 2.0) * (double)(1u << 30));
    for (auto cfg : std::vector<std::pair<int, int>>{{1, 1}, {1, 4}, {1, 16}, {4, 1}, {4, 4}, {8, 2}}) {
        const int F = cfg.first, T = cfg.second;
        std::vector<int> fds;
        for (int f = 0; f < F; ++f) { std::string p = dir + "/pwrite_rate_" + std::to_string(f) + ".bin"; unlink(p.c_str()); fds.push_back(open(p.c_str(), O_WRONLY | O_CREAT | O_TRUNC, 0644)); }
        const size_t n_pieces = per_file / piece;
        std::vector<std::atomic<size_t>> next(F);
        for (auto& x : next) x = 0;
        const double t0 = now();
        std::vector<std::thread> th;
        for (int f = 0; f < F; ++f) for (int t = 0; t < T; ++t)
            th.emplace_back([&, f]() { for (size_t i = next[f].fetch_add(1); i < n_pieces; i = next[f].fetch_add(1)) if (pwrite(fds[f], buf.data(), piece, (off_t)(i * piece)) != (ssize_t)piece) { perror("pwrite"); exit(1); } });
        for (auto& x : th) x.join();
        const double t1 = now();
        for (int fd : fds) close(fd);
        printf("%d file(s) x %d thread(s): %.2f GB in %.2f s = %.2f GB/s\n", F, T, F * per_file / 1e9, t1 - t0, F * per_file / 1e9 / (t1 - t0));
        fflush(stdout);
        for (int f = 0; f < F; ++f) { std::string p = dir + "/pwrite_rate_" + std::to_string(f) + ".bin"; unlink(p.c_str()); }
    }
    return 0;
}
