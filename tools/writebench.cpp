// tools/writebench.cpp -- how fast can result text reach a regular file?  (development aid for dpx_main's printer)
// write() of 8-MiB blocks vs ftruncate + mmap(MAP_SHARED) + memcpy from T threads, fresh file each time.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fcntl.h>
#include <sys/mman.h>
#include <thread>
#include <unistd.h>
#include <vector>
static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
int main(int argc, char **argv) {
    const char *path = argc > 1 ? argv[1] : "/tmp/writebench.out";
    const size_t block = 8u << 20, blocks = 5;
    char *src = (char *)aligned_alloc(4096, block);
    memset(src, 'A', block);
    for (int rep = 0; rep < 3; rep++) {
        unlink(path);
        int fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644);
        double t0 = now();
        for (size_t b = 0; b < blocks; b++) { size_t done = 0; while (done < block) done += (size_t)write(fd, src + done, block - done); }
        double t1 = now();
        close(fd);
        printf("write(): %zu MiB in %.2f ms = %.1f GB/s\n", blocks * block >> 20, (t1 - t0) * 1e3, blocks * block / (t1 - t0) / 1e9);
        for (int T : {1, 2, 4, 8}) {
            unlink(path);
            fd = open(path, O_CREAT | O_RDWR | O_TRUNC, 0644);
            t0 = now();
            size_t pos = 0;
            for (size_t b = 0; b < blocks; b++) {
                if (ftruncate(fd, (off_t)(pos + block)) != 0) { perror("ftruncate"); return 1; }
                char *dst = (char *)mmap(nullptr, block, PROT_READ | PROT_WRITE, MAP_SHARED, fd, (off_t)pos);
                if (dst == MAP_FAILED) { perror("mmap"); return 1; }
                std::vector<std::thread> th;
                const size_t per = block / (size_t)T;
                for (int t = 0; t < T; t++) th.emplace_back([=]() { memcpy(dst + (size_t)t * per, src + (size_t)t * per, per); });
                for (auto &x : th) x.join();
                munmap(dst, block);
                pos += block;
            }
            t1 = now();
            close(fd);
            printf("mmap + memcpy, %d threads: %.2f ms = %.1f GB/s\n", T, (t1 - t0) * 1e3, blocks * block / (t1 - t0) / 1e9);
        }
        // parallel pwrite at disjoint offsets
        for (int T : {2, 4}) {
            unlink(path);
            fd = open(path, O_CREAT | O_WRONLY | O_TRUNC, 0644);
            t0 = now();
            size_t pos = 0;
            for (size_t b = 0; b < blocks; b++) {
                std::vector<std::thread> th;
                const size_t per = block / (size_t)T;
                for (int t = 0; t < T; t++) th.emplace_back([=]() { size_t done = 0; while (done < per) done += (size_t)pwrite(fd, src + (size_t)t * per + done, per - done, (off_t)(pos + (size_t)t * per + done)); });
                for (auto &x : th) x.join();
                pos += block;
            }
            t1 = now();
            close(fd);
            printf("pwrite, %d threads: %.2f ms = %.1f GB/s\n", T, (t1 - t0) * 1e3, blocks * block / (t1 - t0) / 1e9);
        }
    }
    return 0;
}
