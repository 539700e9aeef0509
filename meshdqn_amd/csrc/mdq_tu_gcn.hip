// Translation unit: Q-network forward + learning step (the training kernel re-uses the inference levels).
#include "mdq_gcn.hip"
#include "mdq_gcn_train.hip"
