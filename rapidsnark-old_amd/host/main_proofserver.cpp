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
// 30 s), so parsing a body, answering status polls and enqueueing jobs happen on different cores.  Bodies up to 128000000
// bytes (main_proofserver.cpp:32).
#include <arpa/inet.h>
#include <cerrno>
#include <csignal>
#include <cstring>
#include <iostream>
#include <netinet/in.h>
#include <string>
#include <fcntl.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <ctime>
#include <sys/socket.h>
#include <sys/time.h>
#include <thread>
#include <unistd.h>
#include <vector>

#include "fullprover.hpp"

static const size_t kMaxRequest = 128000000;

static bool send_all(int fd, const std::string &s) {
    size_t off = 0;
    while (off < s.size()) {
        ssize_t k = ::send(fd, s.data() + off, s.size() - off, MSG_NOSIGNAL);
        if (k < 0 && errno == EINTR) continue;
        if (k <= 0) return false;
        off += (size_t)k;
    }
    return true;
}
// recv that is not fooled by a signal landing on this thread (the GPU runtime's threads share the process)
static ssize_t recv_some(int fd, char *buf, size_t len) {
    for (;;) {
        ssize_t k = ::recv(fd, buf, len, 0);
        if (k < 0 && errno == EINTR) continue;
        return k;
    }
}

// per connection: does the client want it kept open after this response?
static thread_local bool t_keep = false;
// a request was refused before its body was read (413, 431, 501, 400): the client may still be sending
static thread_local bool t_unread = false;

static void respond(int fd, int code, const char *reason, const std::string &body, const char *ctype) {
    if (code >= 400 && code != 404 && code != 503) {      // malformed / oversized requests end the connection
        t_keep = false;
        t_unread = true;
    }
    std::string h = "HTTP/1.1 " + std::to_string(code) + " " + reason + "\r\n";
    if (ctype) h += std::string("Content-Type: ") + ctype + "\r\n";
    h += "Content-Length: " + std::to_string(body.size()) + (t_keep ? "\r\nConnection: keep-alive\r\n\r\n" : "\r\nConnection: close\r\n\r\n");
    if (!send_all(fd, h + body)) t_keep = false;
}

// Closing a socket with unread data in its receive buffer sends a reset, and the reset can overtake the response the client
// has not read yet (it is still busy sending the body that was refused).  So: send side shut down, what arrives is read and
// dropped until the client closes or a second has passed, then the socket is closed.
static void lingering_close(int fd) {
    ::shutdown(fd, SHUT_WR);
    timeval brief{0, 200000};
    setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &brief, sizeof brief);
    char sink[65536];
    size_t dropped = 0;
    for (int rounds = 0; rounds < 5 && dropped < ((size_t)8 << 20); rounds++) {
        ssize_t k = recv_some(fd, sink, sizeof sink);
        if (k == 0) break;
        if (k > 0) {
            dropped += (size_t)k;
            rounds = 0;
        }
    }
    ::close(fd);
}

static std::string lower(std::string s) {
    for (auto &c : s) c = (char)tolower((unsigned char)c);
    return s;
}

// One request of a connection.  `buf` carries bytes already received beyond the previous request (a client may send the
// next request before it has read the answer).  Sets t_keep; a closed / idle / broken connection clears it.
static void handle(int fd, FullProver &fp, std::string &buf) {
    t_keep = false;
    t_unread = false;
    size_t hdr_end = buf.find("\r\n\r\n");
    char tmp[65536];
    while (hdr_end == std::string::npos) {
        ssize_t k = recv_some(fd, tmp, sizeof tmp);
        if (k <= 0) return;                      // closed, or idle for longer than the receive timeout
        buf.append(tmp, (size_t)k);
        hdr_end = buf.find("\r\n\r\n");
        if (hdr_end == std::string::npos && buf.size() > 65536) return respond(fd, 431, "Request Header Fields Too Large", "", nullptr);
    }
    std::string head = buf.substr(0, hdr_end);
    size_t le = head.find("\r\n");
    std::string reqline = head.substr(0, le);
    size_t s1 = reqline.find(' '), s2 = reqline.rfind(' ');
    if (s1 == std::string::npos || s2 == s1) return respond(fd, 400, "Bad Request", "", nullptr);
    std::string method = reqline.substr(0, s1), target = reqline.substr(s1 + 1, s2 - s1 - 1);
    size_t qm = target.find('?');
    if (qm != std::string::npos) target.resize(qm);

    size_t clen = 0;
    bool expect100 = false, chunked = false;
    const bool http11 = reqline.size() >= 8 && reqline.compare(reqline.size() - 8, 8, "HTTP/1.1") == 0;
    bool keep = http11;                          // HTTP/1.1: persistent unless the client says close; 1.0: the other way round
    size_t pos = le == std::string::npos ? head.size() : le + 2;
    while (pos < head.size()) {
        size_t e = head.find("\r\n", pos);
        if (e == std::string::npos) e = head.size();
        std::string line = head.substr(pos, e - pos);
        size_t c = line.find(':');
        if (c != std::string::npos) {
            std::string key = lower(line.substr(0, c)), val = line.substr(c + 1);
            while (!val.empty() && val[0] == ' ') val.erase(0, 1);
            if (key == "content-length") clen = (size_t)strtoull(val.c_str(), nullptr, 10);
            if (key == "expect" && lower(val) == "100-continue") expect100 = true;
            if (key == "transfer-encoding" && lower(val) != "identity") chunked = true;
            if (key == "connection") keep = lower(val) == "keep-alive" ? true : (lower(val) == "close" ? false : keep);
        }
        pos = e + 2;
    }
    if (chunked) return respond(fd, 501, "Not Implemented", "Transfer-Encoding is not supported: send Content-Length", "text/plain");   // (closes: the body cannot be skipped)
    if (clen > kMaxRequest) return respond(fd, 413, "Request Entity Too Large", "", nullptr);
    std::string body = buf.substr(hdr_end + 4);
    if (expect100 && body.size() < clen) send_all(fd, "HTTP/1.1 100 Continue\r\n\r\n");
    while (body.size() < clen) {
        ssize_t k = recv_some(fd, tmp, sizeof tmp);
        if (k <= 0) return;
        body.append(tmp, (size_t)k);
    }
    buf = body.size() > clen ? body.substr(clen) : std::string();      // the start of the next request, if any
    body.resize(clen);
    t_keep = keep;

    if (method == "GET" && target == "/status") return respond(fd, 200, "OK", fp.getStatus(), "application/json");
    if (method == "GET" && fp.queueMode() && target.rfind("/status/", 0) == 0 && target.size() > 8) {   // throughput mode: one job's document
        char *end = nullptr;
        const unsigned long long id = strtoull(target.c_str() + 8, &end, 10);
        if (*end) return respond(fd, 404, "Not Found", "Could not find a matching route", "text/plain");
        return respond(fd, 200, "OK", fp.getStatus(id), "application/json");
    }
    if (method == "POST" && (target == "/start" || target == "/stop")) return respond(fd, 200, "OK", "", nullptr);
    if (method == "POST" && target == "/cancel") {
        fp.abort();
        return respond(fd, 200, "OK", "", nullptr);
    }
    if (method == "POST" && fp.queueMode() && target.rfind("/witness/", 0) == 0 && target.size() > 9 && target.find('/', 9) == std::string::npos) {
        uint64_t id = 0;                         // the witness itself (.wtns image): no generator process, no files
        if (!fp.enqueueWitness(std::move(body), target.substr(9), id)) return respond(fd, 503, "Service Unavailable", "{\"error\":\"queue full\"}", "application/json");
        return respond(fd, 200, "OK", "{\"job\":" + std::to_string(id) + "}", "application/json");
    }
    if (method == "POST" && target.rfind("/input/", 0) == 0 && target.size() > 7 && target.find('/', 7) == std::string::npos) {
        if (fp.queueMode()) {      // ZKHIP_QUEUE=n: requests queue up instead of replacing each other
            uint64_t id = 0;
            if (!fp.enqueue(std::move(body), target.substr(7), id)) return respond(fd, 503, "Service Unavailable", "{\"error\":\"queue full\"}", "application/json");
            return respond(fd, 200, "OK", "{\"job\":" + std::to_string(id) + "}", "application/json");
        }
        fp.startProve(std::move(body), target.substr(7));
        return respond(fd, 200, "OK", "", nullptr);
    }
    respond(fd, 404, "Not Found", "Could not find a matching route", "text/plain");
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
        // Every worker polls the listening socket and the connections IT accepted: a kept-alive connection that is silent
        // costs a poll slot, not a thread (16 persistent clients on 8 workers used to wait for each other's idle timeouts).
        {
            int fl = fcntl(ls, F_GETFL, 0);
            fcntl(ls, F_SETFL, fl | O_NONBLOCK);          // a connection another worker took first: EAGAIN, not a blocked thread
        }
        auto worker = [&] {
            struct Conn {
                int fd;
                std::string carry;
                time_t last;
            };
            std::vector<Conn> conns;
            std::vector<pollfd> pfds;
            for (;;) {
                pfds.clear();
                pfds.push_back(pollfd{ls, POLLIN, 0});
                for (auto &c : conns) pfds.push_back(pollfd{c.fd, POLLIN, 0});
                const int pr = ::poll(pfds.data(), (nfds_t)pfds.size(), conns.empty() ? -1 : 1000);
                if (pr < 0 && errno != EINTR) {
                    std::cerr << "poll: " << strerror(errno) << '\n';
                    return;
                }
                const time_t now = time(nullptr);
                // serve what is readable (one request per turn and connection; requests already buffered follow at once)
                for (size_t i = 0; i < conns.size();) {
                    Conn &c = conns[i];
                    const bool readable = pr > 0 && (pfds[i + 1].revents & (POLLIN | POLLHUP | POLLERR));
                    bool keep = true;
                    if (readable || !c.carry.empty()) {
                        do {
                            try {
                                handle(c.fd, fullProver, c.carry);
                            } catch (std::exception &e) {     // (out of memory for a body, a failing file write): this request fails, the server stays
                                t_keep = false;
                                respond(c.fd, 500, "Internal Server Error", e.what(), "text/plain");
                            }
                            keep = t_keep;
                        } while (keep && c.carry.find("\r\n\r\n") != std::string::npos);
                        c.last = now;
                    } else if (now - c.last > 30) {
                        keep = false;                     // idle for half a minute
                    }
                    if (!keep) {
                        if (t_unread) lingering_close(c.fd);
                        else ::close(c.fd);
                        t_unread = false;
                        conns[i] = std::move(conns.back());
                        conns.pop_back();
                        if (i + 1 < pfds.size()) pfds[i + 1] = pfds.back();
                        pfds.pop_back();
                    } else {
                        i++;
                    }
                }
                if (pr > 0 && (pfds[0].revents & POLLIN)) {
                    for (int burst = 0; burst < 16; burst++) {
                        int fd = ::accept(ls, nullptr, nullptr);
                        if (fd < 0) break;               // EAGAIN: somebody else has it (or nothing left)
                        int on = 1;
                        setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &on, sizeof on);
                        timeval slow{5, 0};              // bounds a client that stops in the middle of a request
                        setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &slow, sizeof slow);
                        conns.push_back(Conn{fd, std::string(), now});
                    }
                }
            }
        };
        std::vector<std::thread> pool;
        for (size_t i = 1; i < nthreads; i++) pool.emplace_back(worker);
        worker();
        for (auto &t : pool) t.join();
    } catch (std::exception &e) {
        std::cerr << e.what() << '\n';
        return -1;
    }
}
