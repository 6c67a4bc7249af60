// The REST front end of proverServer (rapidsnark-old_amd/host/http_front.hpp) behind a trivial handler, so that its socket
// behaviour can be tested without a GPU (tests/test_http_front.py):
//   GET /status -> {"status":"ok"}    POST /echo -> the body's length    anything else -> 404
//   g++ -O2 -std=c++17 -pthread -I rapidsnark-old_amd/host tools/http_front_echo.cpp -o tools/http_front_echo
//   tools/http_front_echo <port> [threads=2] [max_body=128000000]
#include <arpa/inet.h>
#include <csignal>
#include "http_front.hpp"

int main(int argc, char **argv) {
    const int port = argc > 1 ? atoi(argv[1]) : 8089;
    const size_t threads = argc > 2 ? (size_t)atoi(argv[2]) : 2, max_body = argc > 3 ? (size_t)atoll(argv[3]) : 128000000;
    signal(SIGPIPE, SIG_IGN);
    int ls = ::socket(AF_INET, SOCK_STREAM, 0), one = 1;
    setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
    sockaddr_in addr{};
    addr.sin_family = AF_INET;
    addr.sin_addr.s_addr = htonl(INADDR_LOOPBACK);
    addr.sin_port = htons((uint16_t)port);
    if (::bind(ls, (sockaddr *)&addr, sizeof addr) < 0 || ::listen(ls, 1024) < 0) {
        perror("bind/listen");
        return 1;
    }
    std::cerr << "ready\n";
    httpfront::serve(ls, threads, max_body, [](httpfront::Request &&rq) {
        httpfront::Response r;
        if (rq.method == "GET" && rq.target == "/status") {
            r.body = "{\"status\":\"ok\"}";
            r.ctype = "application/json";
        } else if (rq.method == "POST" && rq.target == "/echo") {
            r.body = std::to_string(rq.body.size());
            r.ctype = "text/plain";
        } else if (rq.method == "POST" && rq.target == "/throw") {
            throw std::runtime_error("handler failed");
        } else {
            r.code = 404;
            r.reason = "Not Found";
            r.body = "Could not find a matching route";
            r.ctype = "text/plain";
        }
        return r;
    });
}
