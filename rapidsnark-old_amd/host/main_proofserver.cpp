// proverServer <port> <circuit1.zkey> ... <circuitN.zkey>
// The REST shell of the reference (src/main_proofserver.cpp:11-45, src/proverapi.cpp:9-41) over a self-contained
// HTTP/1.1 front end (the reference's Pistache is an empty submodule):
//   GET  /status            -> FullProver::getStatus() document, application/json
//   POST /input/:circuit    -> body = circom input JSON; 200 at once, job runs in the background
//   POST /cancel            -> abort()
//   POST /start, /stop      -> 200, no-ops (proverapi.cpp:27-33)
// Throughput mode (ZKHIP_QUEUE=n, see fullprover.hpp): POST /input/:circuit answers {"job":id} (503 when n requests
// are already waiting), GET /status/<id> reports that job, and POST /witness/:circuit takes the witness itself
// (a .wtns image as the body) instead of running the generator; everything else is unchanged.
// The reference runs ONE HTTP thread (Http::Endpoint::options().threads(1), main_proofserver.cpp:34) and one request
// per connection; behind eight GPUs that loop is the bound (round-2 measurement: 2018 proofs/s through REST against
// 3300 through the C-ABI at 2^14).  Here ZKHIP_HTTP_THREADS workers (default 8) each poll the listening socket and
// the connections they accepted, with keep-alive (any number of persistent connections; idle ones are dropped after
// 30 s), so parsing a body, answering status polls and enqueueing jobs happen on different cores; sockets are non-blocking
// with per-connection parse state (http_front.hpp), so one slow upload never stalls a worker's other connections.
// Bodies up to 128000000 bytes (main_proofserver.cpp:32).
#include <arpa/inet.h>
#include <cerrno>
#include <csignal>
#include <cstring>
#include <iostream>
#include <netinet/in.h>
#include <string>
#include <sys/socket.h>
#include <unistd.h>

#include "fullprover.hpp"
#include "http_front.hpp"

static const size_t kMaxRequest = 128000000;

// One complete request -> its answer (src/proverapi.cpp:9-41 + the throughput-mode routes).  Runs on an HTTP worker; the
// sockets, keep-alive, framing and deadlines are http_front.hpp's.
static httpfront::Response route(FullProver &fp, httpfront::Request &&rq) {
    using R = httpfront::Response;
    auto make = [](int code, const char *reason, std::string body, const char *ctype) {
        R r;
        r.code = code;
        r.reason = reason;
        r.body = std::move(body);
        r.ctype = ctype;
        return r;
    };
    const std::string &method = rq.method, &target = rq.target;
    if (method == "GET" && target == "/status") return make(200, "OK", fp.getStatus(), "application/json");
    if (method == "GET" && fp.queueMode() && target.rfind("/status/", 0) == 0 && target.size() > 8) {   // throughput mode: one job's document
        char *end = nullptr;
        const unsigned long long id = strtoull(target.c_str() + 8, &end, 10);
        if (*end) return make(404, "Not Found", "Could not find a matching route", "text/plain");
        return make(200, "OK", fp.getStatus(id), "application/json");
    }
    if (method == "POST" && (target == "/start" || target == "/stop")) return make(200, "OK", "", nullptr);
    if (method == "POST" && target == "/cancel") {
        fp.abort();
        return make(200, "OK", "", nullptr);
    }
    if (method == "POST" && fp.queueMode() && target.rfind("/witness/", 0) == 0 && target.size() > 9 && target.find('/', 9) == std::string::npos) {
        uint64_t id = 0;                         // the witness itself (.wtns image): no generator process, no files
        if (!fp.enqueueWitness(std::move(rq.body), target.substr(9), id)) return make(503, "Service Unavailable", "{\"error\":\"queue full\"}", "application/json");
        return make(200, "OK", "{\"job\":" + std::to_string(id) + "}", "application/json");
    }
    if (method == "POST" && target.rfind("/input/", 0) == 0 && target.size() > 7 && target.find('/', 7) == std::string::npos) {
        if (fp.queueMode()) {      // ZKHIP_QUEUE=n: requests queue up instead of replacing each other
            uint64_t id = 0;
            if (!fp.enqueue(std::move(rq.body), target.substr(7), id)) return make(503, "Service Unavailable", "{\"error\":\"queue full\"}", "application/json");
            return make(200, "OK", "{\"job\":" + std::to_string(id) + "}", "application/json");
        }
        fp.startProve(std::move(rq.body), target.substr(7));
        return make(200, "OK", "", nullptr);
    }
    return make(404, "Not Found", "Could not find a matching route", "text/plain");
}

int main(int argc, char **argv) {
    if (argc < 3) {
        std::cerr << "Invalid number of parameters:\n";
        std::cerr << "Usage: proverServer <port> <circuit1.zkey> <circuit2.zkey> ... <circuitN.zkey> \n";
        return -1;
    }
    try {
        std::cerr << "Initializing server...\n";
        int port = std::stoi(argv[1]);
        std::string *zkeyFileNames = new std::string[argc - 2];
        for (int i = 0; i < argc - 2; i++) zkeyFileNames[i] = argv[i + 2];
        FullProver fullProver(zkeyFileNames, argc - 2);
        delete[] zkeyFileNames;

        signal(SIGPIPE, SIG_IGN);
        int ls = ::socket(AF_INET, SOCK_STREAM, 0);
        if (ls < 0) throw std::runtime_error(std::string("socket: ") + strerror(errno));
        int one = 1;
        setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
        sockaddr_in addr{};
        addr.sin_family = AF_INET;
        addr.sin_addr.s_addr = htonl(INADDR_ANY);
        addr.sin_port = htons((uint16_t)port);
        if (::bind(ls, (sockaddr *)&addr, sizeof addr) < 0) throw std::runtime_error(std::string("bind: ") + strerror(errno));
        if (::listen(ls, 1024) < 0) throw std::runtime_error(std::string("listen: ") + strerror(errno));
        size_t nthreads = 8;
        if (const char *e = getenv("ZKHIP_HTTP_THREADS")) nthreads = (size_t)strtoul(e, nullptr, 10);
        if (nthreads < 1) nthreads = 1;
        if (nthreads > 256) nthreads = 256;
        std::cerr << "Server ready on port " << port << "...\n";
        httpfront::serve(ls, nthreads, kMaxRequest, [&](httpfront::Request &&rq) { return route(fullProver, std::move(rq)); });
    } catch (std::exception &e) {
        std::cerr << e.what() << '\n';
        return -1;
    }
}
