#include "mp2p_hip_host.hpp"
#include <cstdio>
using namespace mp2p_hip_host;
int main(){
  std::vector<mp2p_hip_pair_pt2pt> v(1000);
  for (size_t i=0;i<v.size();i++){ v[i].globalIdx=i*7+1; v[i].localIdx=i; v[i].errorSquareAfterTransformation=0.5f*i; }
  ListPrint a = list_print(v.data(), v.size());
  int bad=0;
  for (size_t cut : {0ul,1ul,5ul,31ul,32ul,33ul,63ul,64ul,65ul,100ul,128ul,999ul,1000ul}) {
    ListPrint f; print_feed(f, v.data(), cut); print_feed(f, v.data()+cut, v.size()-cut);
    if (!(f==a)) { printf("chunk mismatch at %zu\n", cut); bad++; }
  }
  // brute-force definition
  ListPrint b; for (size_t i=0;i<v.size();i++) if (sampled_pos(i)) b.h = mix64(b.h, pair_word(v[i])); b.last=pair_word(v.back()); b.n=v.size();
  if (!(b==a)) { printf("definition mismatch\n"); bad++; }
  auto w=v; std::swap(w[64],w[65]); if (list_print(w.data(),w.size())==a) { printf("swap undetected\n"); bad++; }
  printf(bad? "FAIL\n":"ok\n"); return bad;
}
