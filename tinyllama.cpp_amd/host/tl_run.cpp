// tl_run — minimal command-line driver of the C++ host layer (the reference's main.cpp drives TinyLlamaSession).
//   tl_run model.gguf --ids 1,15,27 [--steps 32] [--temperature 0] [--top-k 40] [--top-p 0.9] [--penalty 1.3]
//          [--seed 0] [--ctx 2048] [--device 0]
//   tl_run model.gguf --prompt "text" [--system "text"] [--qa 1] [...]      (needs the file's tokenizer metadata)
// Prints one JSON line: {"generated": [...], "load_s": ..., "gen_s": ..., "tokens_per_s": ...}; with --prompt the
// generated text follows on the next lines (the prompt goes through format_prompt / encode as in api.cpp:505-575).
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "tl_session.hpp"

int main(int argc, char** argv) {
  if (argc < 2) {
    fprintf(stderr, "usage: %s model.gguf (--ids a,b,c | --prompt TEXT [--system TEXT] [--qa 1]) [--steps N] [--temperature T] [--top-k K] [--top-p P] "
                    "[--penalty R] [--seed S] [--ctx C] [--device D]\n", argv[0]);
    return 2;
  }
  std::string path = argv[1];
  std::vector<int> ids;
  int steps = 32, top_k = 40, ctx = 0, device = 0;
  float temperature = 0.0f, top_p = 0.9f, penalty = 1.3f;
  long seed = -1;
  std::string prompt, system_prompt;
  bool have_prompt = false, qa = false;
  for (int i = 2; i + 1 < argc; i += 2) {
    const std::string k = argv[i];
    const char* v = argv[i + 1];
    if (k == "--ids") {
      for (const char* p = v; *p;) {
        char* end = nullptr;
        ids.push_back((int)strtol(p, &end, 10));
        if (end == p) break;
        p = *end == ',' ? end + 1 : end;
      }
    } else if (k == "--prompt") {
      prompt = v;
      have_prompt = true;
    } else if (k == "--system") system_prompt = v;
    else if (k == "--qa") qa = atoi(v) != 0;
    else if (k == "--steps") steps = atoi(v);
    else if (k == "--temperature") temperature = (float)atof(v);
    else if (k == "--top-k") top_k = atoi(v);
    else if (k == "--top-p") top_p = (float)atof(v);
    else if (k == "--penalty") penalty = (float)atof(v);
    else if (k == "--seed") seed = atol(v);
    else if (k == "--ctx") ctx = atoi(v);
    else if (k == "--device") device = atoi(v);
    else {
      fprintf(stderr, "unknown option %s\n", k.c_str());
      return 2;
    }
  }
  try {
    tlhost::Session session(path, device, ctx);
    session.set_repetition_penalty(penalty);
    if (seed >= 0) session.seed((uint32_t)seed);
    if (have_prompt) {
      const tlhost::Tokenizer* tok = session.get_tokenizer();
      if (!tok) throw std::runtime_error("Model or tokenizer not loaded.");
      ids = tok->encode(tok->format_prompt(prompt, system_prompt, qa), true, false);
    }
    const auto t0 = std::chrono::steady_clock::now();
    const std::vector<int> out = session.generate(ids, steps, temperature, top_k, top_p);
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    printf("{\"generated\": [");
    for (size_t i = 0; i < out.size(); ++i) printf("%s%d", i ? ", " : "", out[i]);
    printf("], \"load_s\": %.3f, \"gen_s\": %.4f, \"tokens_per_s\": %.1f}\n", session.load_stats().seconds, dt,
           out.empty() ? 0.0 : out.size() / dt);
    if (have_prompt) printf("%s\n", session.get_tokenizer()->decode(out, true).c_str());
  } catch (const std::exception& e) {
    fprintf(stderr, "tl_run: %s\n", e.what());
    return 1;
  }
  return 0;
}
