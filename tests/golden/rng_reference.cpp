// Generator of tests/golden/rng_golden.json.  Uses libstdc++'s own <random> with the
// constructs the reference uses (host/RunHardware.cpp:31-35, test/TestSimulation.cpp:46-55):
//   std::default_random_engine rng(kSeed = 5);
//   std::uniform_real_distribution<double> / std::uniform_int_distribution<unsigned long> dist(1, 10);
// Prints JSON; doubles as C99 hex floats so the fixture is bit-exact.
#include <cstdio>
#include <random>
int main() {
  const int count = 256;
  {
    std::default_random_engine rng(5);
    std::uniform_real_distribution<double> dist(1, 10);
    std::printf("{\n  \"seed\": 5,\n  \"real_hex\": [");
    for (int i = 0; i < count; ++i) std::printf("%s\"%a\"", i ? ", " : "", dist(rng));
    std::printf("],\n");
  }
  {
    std::default_random_engine rng(5);
    std::uniform_real_distribution<double> dist(1, 10);
    std::printf("  \"real_first4_decimal\": [");
    for (int i = 0; i < 4; ++i) std::printf("%s\"%.17g\"", i ? ", " : "", dist(rng));
    std::printf("],\n");
  }
  {
    std::default_random_engine rng(5);
    std::uniform_int_distribution<unsigned long> dist(1, 10);
    std::printf("  \"int\": [");
    for (int i = 0; i < count; ++i) std::printf("%s%lu", i ? ", " : "", dist(rng));
    std::printf("]\n}\n");
  }
  return 0;
}
