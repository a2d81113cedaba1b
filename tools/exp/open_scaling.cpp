#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <thread>
#include <vector>
#include <atomic>
#include <chrono>
#include <string>
#include <fcntl.h>
#include <unistd.h>
#include <sys/stat.h>
#include <zlib.h>
#include <cstring>
int main(int argc, char** argv) {
    int T = argc > 1 ? atoi(argv[1]) : 8, mode = argc > 2 ? atoi(argv[2]) : 0, n = 10000;
    std::vector<std::string> paths(n);
    for (int i = 0; i < n; ++i) { char b[64]; snprintf(b, 64, "/dev/shm/rdtest/%05d.gz", i); paths[i] = b; }
    for (int rep = 0; rep < 3; ++rep) {
        auto t0 = std::chrono::steady_clock::now();
        std::atomic<int> next(0);
        std::atomic<uint64_t> tot(0);
        auto work = [&]() {
            uint8_t buf[8192], out[16384]; uint64_t mine = 0;
            z_stream zs; memset(&zs, 0, sizeof zs); inflateInit2(&zs, 15 + 32);
            for (;;) { int i = next.fetch_add(1); if (i >= n) break;
                if (mode == 3) { volatile double x = 0; for (int j = 0; j < 20000; ++j) x += j; continue; }
                int fd = open(paths[i].c_str(), O_RDONLY); 
                if (mode >= 1) { struct stat st; fstat(fd, &st); }
                ssize_t r = read(fd, buf, sizeof buf); close(fd); mine += r;
                if (mode >= 2) { inflateReset2(&zs, 15 + 32); zs.next_in = buf; zs.avail_in = r; zs.next_out = out; zs.avail_out = sizeof out; inflate(&zs, Z_NO_FLUSH); mine += zs.total_out; }
            }
            inflateEnd(&zs);
            tot += mine; };
        std::vector<std::thread> pool;
        for (int w = 1; w < T; ++w) pool.emplace_back(work);
        work();
        for (auto& t : pool) t.join();
        double s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        printf("T=%d mode=%d %.4f s  (%.1f us per file-thread) %lu\n", T, mode, s, s * T / n * 1e6, (unsigned long)tot.load());
    }
}
