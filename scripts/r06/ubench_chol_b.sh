cd scripts/ubench
for b in f64_base f64_vgpr f64_vgpr_bc1 f64_vgpr_bc2; do echo "== $b"; timeout 60 ./$b 1 | tail -2; done
for b in chol_occ1 chol_occ2 chol_occ1_vgpr chol_occ2_vgpr; do for args in "1202 384 288" "3200" "6002"; do echo "== $b $args"; timeout 60 ./$b $args | grep -v "mode=1"; done; done
