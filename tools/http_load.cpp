// Request rate of proverServer's REST front end: T threads, one keep-alive connection each, GET <path> back to back.
//   g++ -O2 -std=c++17 -pthread tools/http_load.cpp -o tools/http_load && tools/http_load <port> <threads> <seconds> [path=/status]
// (tools/, not product code.)
#include <arpa/inet.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <string>
#include <sys/socket.h>
#include <thread>
#include <unistd.h>
#include <vector>

static int dial(int port) {
    int fd = socket(AF_INET, SOCK_STREAM, 0);
    sockaddr_in a{};
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    a.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    if (connect(fd, (sockaddr *)&a, sizeof a) != 0) { perror("connect"); exit(1); }
    int on = 1;
    setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &on, sizeof on);
    return fd;
}

static long client(int port, const std::string &req, std::atomic<bool> &stop) {
    int fd = dial(port);
    std::string buf;
    char tmp[8192];
    long n = 0;
    while (!stop.load(std::memory_order_relaxed)) {
        if (send(fd, req.data(), req.size(), MSG_NOSIGNAL) != (ssize_t)req.size()) break;
        size_t he;                           // one response: headers + Content-Length bytes
        bool dead = false;
        while ((he = buf.find("\r\n\r\n")) == std::string::npos) {
            ssize_t k = recv(fd, tmp, sizeof tmp, 0);
            if (k <= 0) { dead = true; break; }
            buf.append(tmp, (size_t)k);
        }
        if (dead) break;
        size_t cl = buf.find("Content-Length: ");
        size_t len = cl == std::string::npos ? 0 : strtoul(buf.c_str() + cl + 16, nullptr, 10);
        while (buf.size() < he + 4 + len) {
            ssize_t k = recv(fd, tmp, sizeof tmp, 0);
            if (k <= 0) { dead = true; break; }
            buf.append(tmp, (size_t)k);
        }
        if (dead) break;
        buf.erase(0, he + 4 + len);
        n++;
    }
    close(fd);
    return n;
}

int main(int argc, char **argv) {
    if (argc < 4) { fprintf(stderr, "usage: http_load <port> <threads> <seconds> [path]\n"); return 2; }
    const int port = atoi(argv[1]), nt = atoi(argv[2]);
    const double secs = atof(argv[3]);
    const std::string path = argc > 4 ? argv[4] : "/status";
    const std::string req = "GET " + path + " HTTP/1.1\r\nHost: localhost\r\n\r\n";
    std::atomic<long> total{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int t = 0; t < nt; t++) th.emplace_back([&] { total += client(port, req, stop); });
    auto t0 = std::chrono::steady_clock::now();
    std::this_thread::sleep_for(std::chrono::duration<double>(secs));
    stop = true;
    for (auto &t : th) t.join();
    double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"threads\": %d, \"seconds\": %.2f, \"requests\": %ld, \"requests_per_s\": %.0f, \"path\": \"%s\"}\n", nt, dt, total.load(), total.load() / dt, path.c_str());
    return 0;
}
