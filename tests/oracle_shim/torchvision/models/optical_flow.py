raft_large = None
Raft_Large_Weights = None
