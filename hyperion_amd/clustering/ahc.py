import numpy as np

from ..hyp_model import HypModel


class AHC(HypModel):
    """Agglomerative clustering of a score matrix (hyperion/clustering/ahc.py:16-110).  The tree is
    scipy's linkage on the upper triangle, on the host as in the reference: it is sequential
    O(n^2) work on the few thousand segments of one recording.  metric "llr" / "prob": larger
    means closer; anything else is handed to linkage as a distance metric."""

    def __init__(self, method="average", metric="llr", **kwargs):
        super().__init__(**kwargs)
        self.method = method
        self.metric = metric
        self.Z = None
        self.flat_clusters = None

    def fit(self, x, mask=None):
        """x: [n, n] scores (host array or device tensor), upper triangle used; mask False at
        (i, j) forbids merging i and j."""
        from scipy.cluster.hierarchy import linkage

        self.flat_clusters = None
        if hasattr(x, "cpu"):
            x = x.cpu().numpy()
        x = np.asarray(x, dtype=np.float64)
        if mask is not None:
            x = np.where(np.asarray(mask, dtype=bool), x, -1e10)
        pairs = x[np.triu_indices(x.shape[0], k=1)]
        if self.metric == "llr":
            top = np.max(pairs)
            self.Z = linkage(top - pairs, method=self.method)
            self.Z[:, 2] = top - self.Z[:, 2]
        elif self.metric == "prob":
            self.Z = linkage(1 - pairs, method=self.method)
            self.Z[:, 2] = 1 - self.Z[:, 2]
        else:
            self.Z = linkage(pairs, method=self.method, metric=self.metric)

    def get_flat_clusters(self, t, criterion="threshold"):
        if criterion == "threshold":
            return self.get_flat_clusters_from_thr(t)
        return self.get_flat_clusters_from_num_clusters(t)

    def get_flat_clusters_from_thr(self, thr):
        """Apply every merge whose score is at least thr (distance at most thr)."""
        keep = self.Z[:, 2] >= thr if self.metric in ("llr", "prob") else self.Z[:, 2] <= thr
        return self.get_flat_clusters_from_num_clusters(self.Z.shape[0] + 1 - int(np.sum(keep)))

    def get_flat_clusters_from_num_clusters(self, num_clusters):
        """Labels after the first n - num_clusters merges.  A merged cluster is known by the number
        of the merge that made it (n + merge index), a single point by its own index, and labels
        are the ranks of those numbers (ahc.py:81-99)."""
        n = self.Z.shape[0] + 1
        merges = n - min(n, num_clusters)
        members = {i: [i] for i in range(n)}
        for i in range(merges):
            a, b = int(self.Z[i, 0]), int(self.Z[i, 1])
            members[n + i] = members.pop(a) + members.pop(b)
        labels = np.empty(n, dtype=int)
        for rank, node in enumerate(sorted(members)):
            labels[members[node]] = rank
        return labels

    def compute_flat_clusters(self):
        """Labels for every possible number of clusters: row i of `flat_clusters` [n, n] is the
        assignment with n - i clusters, numbered by rank of the cluster's node number as in
        get_flat_clusters_from_num_clusters (ahc.py:115-134)."""
        n = self.Z.shape[0] + 1
        node = np.arange(n)  # node number of the cluster each point is in
        flat = np.zeros((n, n), dtype=int)
        flat[0] = node
        for i in range(n - 1):
            node = np.where((node == self.Z[i, 0]) | (node == self.Z[i, 1]), n + i, node)
            flat[i + 1] = np.unique(node, return_inverse=True)[1]
        self.flat_clusters = flat

    def evaluate_homogeneity_completeness_tradeoff(self, true_labels):
        """(homogeneity, completeness) of every row of `flat_clusters` against the true labels
        (ahc.py:136-158; scikit-learn's scores, on the host as there)."""
        from sklearn.metrics import completeness_score, homogeneity_score

        from ..hyp_defs import float_cpu

        if self.flat_clusters is None:
            self.compute_flat_clusters()
        h = np.array([homogeneity_score(true_labels, row) for row in self.flat_clusters], dtype=float_cpu())
        c = np.array([completeness_score(true_labels, row) for row in self.flat_clusters], dtype=float_cpu())
        return h, c
